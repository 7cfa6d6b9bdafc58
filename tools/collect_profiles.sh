# Round 3: the command list behind profiles/r3_*.  Run on the GPU box: gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
# (rocprofv3 rules on this pool: cd /tmp && export TMPDIR=/tmp first; counters only with --kernel-trace.)
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err)
(timeout 300 python bench.py --workload vlp16 --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 > $O/bench_vlp16.json 2> $O/bench_vlp16.err)
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
(LIO_RESIDENT_MOMENTS=0 LIO_DEBUG_TIMING=1 timeout 200 $B > $O/bench_resident_off.json 2> $O/bench_resident_off.err)
(LIO_DEBUG_TIMING=1 timeout 200 $B > $O/bench_dbg.json 2> $O/bench_dbg.err)
(LIO_RESIDENT_ROUNDS=1 LIO_DEBUG_TIMING=1 timeout 200 $B > $O/bench_resident_rounds.json 2> $O/bench_resident_rounds.err)
(LIO_HOST_SIGNAL=0 timeout 200 $B > $O/bench_no_host_signal.json 2> $O/bench_no_host_signal.err)
(LIO_DEVICE_SOLVE=1 timeout 200 $B --steps 20 > $O/bench_device_solve.json 2> $O/bench_device_solve.err)
(LIO_DEVICE_MARG=1 timeout 200 $B --steps 20 > $O/bench_device_marg.json 2> $O/bench_device_marg.err)
(timeout 300 python tools/stress_determinism.py 2000 > $O/stress.log 2>&1)
(timeout 120 tools/micro/pingpong > $O/pingpong.txt 2>&1)
(LIO_DEBUG_TIMING=1 timeout 120 python profiles/pp_profile.py > $O/pp_dbg.log 2>&1)
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 20 --warmup 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_s/st_results.db > $O/kernel_stats.md
(timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_pp -o pp -- python $R/profiles/pp_profile.py > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_pp/pp_results.db > $O/pp_kernel_stats.md
(timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -o f -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 5 --warmup 1 --windows 0 --keyframes 40 > /dev/null 2>&1)
(timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_w -o w -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 5 --warmup 1 --windows 0 --keyframes 40 > /dev/null 2>&1)
python $R/profiles/pmc_summary.py /tmp/prof_f/f_results.db /tmp/prof_w/w_results.db $O/pmc.json > $O/pmc_hbm_traffic.md
(timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_m -o m -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 5 --warmup 1 --windows 0 --keyframes 0 > /dev/null 2>&1)
(LIO_MOMENTS=mfma timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d /tmp/prof_mb -o mb -- python $R/tools/batched_moments.py 512 > $O/batched_mfma.txt 2>&1)
(LIO_MOMENTS=valu timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d /tmp/prof_vb -o vb -- python $R/tools/batched_moments.py 512 > $O/batched_valu.txt 2>&1)
(LIO_MOMENTS=mfma timeout 120 python $R/tools/batched_moments.py > $O/batched_mfma_plain.txt 2>&1)
(LIO_MOMENTS=valu timeout 120 python $R/tools/batched_moments.py > $O/batched_valu_plain.txt 2>&1)
(timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d /tmp/prof_q -o q -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 5 --warmup 1 --windows 0 --keyframes 40 > /dev/null 2>&1)
python - <<'PY' > $O/pmc_counters.md
import sqlite3, glob
def table(dbglob, title, pick):
    db = glob.glob(dbglob)
    if not db:
        print(f"## {title}\n\n(no database)\n"); return
    cur = sqlite3.connect(db[0]).cursor()
    print(f"## {title}\n\n| kernel | grid | counter | launches | average per launch | avg duration us (profiled) |\n|---|---|---|---|---|---|")
    q = "select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, grid_size, counter_name order by kernel_name, grid_size, counter_name"
    for k, g, c, n, v, d in cur.execute(q):
        if any(t in k for t in pick):
            print(f"| `{k.split('(')[0][:70]}` | {g} | {c} | {n} | {v:.1f} | {(d or 0) / 1e3:.2f} |")
    print()
table("/tmp/prof_m/*results.db", "fp64 MFMA counters, the bench step (resident moments kernel)", ("k_lidar_moments", "k_moment_reduce"))
table("/tmp/prof_mb/*results.db", "batched moments, B = 512, fp64-MFMA form (LIO_MOMENTS=mfma)", ("k_lidar_moments",))
table("/tmp/prof_vb/*results.db", "batched moments, B = 512, structured fp64-VALU form (LIO_MOMENTS=valu)", ("k_lidar_moments",))
table("/tmp/prof_q/*results.db", "SQ counters of the search kernels and the keyframe batch (--keyframes 40)", ("k_odom_round", "k_features", "k_kf_round", "k_odom_update_wide", "k_ring_pick"))
PY
tail -3 $O/pytest_gpu.log
