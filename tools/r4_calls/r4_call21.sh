#!/bin/bash
# Round 4, GPU call 21: stream priorities (estimator stream high, side stream low) A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run prio1 LIO_STREAM_PRIORITY=1
run prio0 LIO_STREAM_PRIORITY=0
run prio1b LIO_STREAM_PRIORITY=1
run prio0b LIO_STREAM_PRIORITY=0
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
