# Round 4, last commit: the command list behind profiles/r4_final_* (gpurun --timeout 1500 -- 'bash tools/r4_final.sh').
# (rocprofv3 rules on this pool: cd /tmp && export TMPDIR=/tmp first; counters only with --kernel-trace.)
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4final
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err)
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
(LIO_SPLIT_FACTOR=0 timeout 200 $B > $O/bench_split0.json 2> $O/bench_split0.err)
(LIO_RESIDENT_ROUNDS=1 timeout 200 $B > $O/bench_resident_rounds.json 2> $O/bench_resident_rounds.err)
(LIO_VOX_FUSED=1 timeout 200 $B > $O/bench_vox_fused.json 2> $O/bench_vox_fused.err)
(LIO_DEBUG_TIMING=1 timeout 200 $B > $O/bench_dbg.json 2> $O/bench_dbg.err)
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 20 --warmup 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_s/st_results.db > $O/kernel_stats.md
(timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_pp -o pp -- python $R/profiles/pp_profile.py > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_pp/pp_results.db > $O/pp_kernel_stats.md
(timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -o f -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 5 --warmup 1 --windows 0 --keyframes 0 > /dev/null 2>&1)
(timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_w -o w -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 5 --warmup 1 --windows 0 --keyframes 0 > /dev/null 2>&1)
python $R/profiles/pmc_summary.py /tmp/prof_f/f_results.db /tmp/prof_w/w_results.db $O/pmc.json > $O/pmc_hbm_traffic.md
(timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_m -o m -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --steps 5 --warmup 1 --windows 0 --keyframes 0 > /dev/null 2>&1)
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
PY
tail -3 $O/pytest_gpu.log
for f in $O/bench.json $O/bench_split0.json $O/bench_resident_rounds.json $O/bench_vox_fused.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
head -8 $O/kernel_stats.md
