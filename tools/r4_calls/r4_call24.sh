#!/bin/bash
# Round 4, GPU call 24: the update kernel's fold with eight loads in flight (same sums) — parity tests, bench, kernel stats.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4x; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py tests/test_gpu_zz_ref_state.py tests/test_gpu_degenerate.py -q -x > $O/pytest_some.log 2>&1; tail -2 $O/pytest_some.log)
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
(timeout 200 $B > $O/bench_a.json 2> $O/bench_a.err); (timeout 200 $B > $O/bench_b.json 2> $O/bench_b.err)
for f in $O/bench_a.json $O/bench_b.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d.get("stages_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o st -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --windows 0 --keyframes 0 --steps 20 --warmup 3 > /dev/null 2>&1)
python $R/profiles/summarize_rocpd.py /tmp/prof_s/st_results.db > $O/kernel_stats.md
grep -E "k_odom_update_wide|k_odom_round|k_features" $O/kernel_stats.md | cut -c1-60,170-
