set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2final
mkdir -p $O
cd $R
(timeout 700 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err)
(LIO_DEVICE_SOLVE=1 LIO_DEBUG_TIMING=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 20 --windows 0 --keyframes 0 > $O/bench_device_solve.json 2> $O/bench_device_solve.err)
(LIO_DEVICE_MARG=1 LIO_DEBUG_TIMING=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 20 --windows 0 --keyframes 0 > $O/bench_device_marg.json 2> $O/bench_device_marg.err)
(LIO_HOST_SIGNAL=0 timeout 200 python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 > $O/bench_no_host_signal.json 2> $O/bench_no_host_signal.err)
(LIO_DEBUG_TIMING=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 10 --windows 0 --keyframes 0 > /dev/null 2> $O/bench_dbg.err)
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-pmc --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_s/st_results.db > $O/kernel_stats.md
(timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -o f -- python $R/bench.py --no-pmc --no-cpu-baseline --steps 5 --warmup 1 --windows 0 --keyframes 40 > /dev/null 2>&1)
(timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_w -o w -- python $R/bench.py --no-pmc --no-cpu-baseline --steps 5 --warmup 1 --windows 0 --keyframes 40 > /dev/null 2>&1)
python $R/profiles/pmc_summary.py /tmp/prof_f/f_results.db /tmp/prof_w/w_results.db $O/pmc.json > $O/pmc_hbm_traffic.md
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/prof_q -o q -- python $R/bench.py --no-pmc --no-cpu-baseline --steps 5 --warmup 1 --windows 0 --keyframes 0 > /dev/null 2>&1)
python - <<'PY' > $O/pmc_sq_after.md
import sqlite3, glob
db = glob.glob("/tmp/prof_q/*results.db")
cur = sqlite3.connect(db[0]).cursor()
print("| kernel | counter | launches | average per launch |")
print("|---|---|---|---|")
for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if any(t in k for t in ("k_odom_round", "k_features", "k_odom_update_wide", "k_lidar_moments(", "k_moment_reduce")):
        print(f"| `{k.split('(')[0][:60]}` | {c} | {n} | {v:.1f} |")
PY
tail -3 $O/pytest_gpu.log
