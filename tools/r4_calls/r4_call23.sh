#!/bin/bash
# Round 4, GPU call 23: the older frames features riding in the launches of the rounds update blocks — parity tests, bench A/B, timeline.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
(timeout 600 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py tests/test_gpu_zz_ref_state.py tests/test_gpu_ref_estimator_steps.py -q -x > $O/pytest_some.log 2>&1; tail -2 $O/pytest_some.log)
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run ride1 LIO_RIDE_FEATURES=1
run ride0 LIO_RIDE_FEATURES=0
run ride1b LIO_RIDE_FEATURES=1
run ride0b LIO_RIDE_FEATURES=0
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_g -o g -- python $R/bench.py --no-pmc --no-cpu-baseline --no-fed --windows 0 --keyframes 0 --steps 10 --warmup 2 > /dev/null 2>&1)
python $R/profiles/gap_summary.py /tmp/prof_g/g_results.db > $O/gaps.md
sed -n 16,34p $O/gaps.md
