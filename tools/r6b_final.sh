# Round 6, second session, last commit: the command list behind profiles/r6b_final_* (gpurun --timeout 2400 -- 'bash tools/r6b_final.sh').
# (rocprofv3 rules on this pool: cd /tmp && export TMPDIR=/tmp first; counters only with --kernel-trace.)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6bfinal
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
SECONDS=0
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err)
echo "bench.py default run: $SECONDS s"
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --windows 0 --keyframes 0 --steps 20 --warmup 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_s/st_results.db > $O/kernel_stats.md
for B in 64 512; do
  (timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -o b -- python $R/tools/batch_profile.py $B 3 > /dev/null 2>&1)
  python $R/profiles/summarize_rocpd.py /tmp/prof_b$B/b_results.db > $O/batch${B}_kernel_stats.md
done
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_f -o f -- python $R/tools/batch_profile.py 64 2 > /dev/null 2>&1)
(timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_w -o w -- python $R/tools/batch_profile.py 64 2 > /dev/null 2>&1)
python $R/profiles/pmc_summary.py /tmp/prof_f/f_results.db /tmp/prof_w/w_results.db $O/batch64_pmc.json > $O/batch64_pmc_hbm_traffic.md 2>&1
(timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_m -o m -- python $R/tools/batch_profile.py 64 2 > /dev/null 2>&1)
python - <<'PY' > $O/batch64_pmc_counters.md
import sqlite3, glob
db = glob.glob("/tmp/prof_m/*results.db")
print("## fp64 MFMA counters, 64 windows per launch chain (tools/batch_profile.py 64 2)\n")
if not db:
    print("(no database)")
else:
    cur = sqlite3.connect(db[0]).cursor()
    print("| kernel | grid | counter | launches | average per launch | avg duration us (profiled) |\n|---|---|---|---|---|---|")
    try:
        q = "select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, grid_size, counter_name order by kernel_name, grid_size, counter_name"
        for k, g, c, n, v, d in cur.execute(q):
            if "k_bw_" in k:
                print(f"| `{k.split('(')[0][:70]}` | {g} | {c} | {n} | {v:.1f} | {(d or 0) / 1e3:.2f} |")
    except Exception as e:
        print("(query failed:", e, ")")
PY
(timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_pp -o pp -- python $R/tools/pp_batch_profile.py 64 8 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_pp/pp_results.db > $O/pp_batch64_kernel_stats.md
cd $R
for B in 1 8 64 256; do timeout 200 python tools/pp_batch_profile.py $B 2>&1 | tail -1; done > $O/pp_batch_profile.txt 2>&1
for B in 64 512 8; do timeout 200 python tools/batch_profile.py $B 8 2>&1 | tail -2 | cut -c1-420; done > $O/batch_profile.txt 2>&1
python - "$O/bench.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bench", d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic")})
    print("cpu_baseline", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("kind"))
    for p in d["batched"]["points"]:
        print("batched", p["windows"], p["value"], p["ms_per_batch_step"], {k.split(" ")[0]: v.get("device_ms") for k, v in p["stages"].items()})
    print("keyframes", d.get("keyframe_batch", {}).get("keyframes_per_s"), "ms_per_scan", d.get("ms_per_scan", {}).get("total"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
head -12 $O/kernel_stats.md | cut -c1-160
head -12 $O/batch64_kernel_stats.md | cut -c1-160
tail -3 $O/bench.err
