#!/bin/bash
# A/B of the resident moments kernel on the GPU box.  Output under gpurun_out/r3_res/.
O=gpurun_out/r3_res; mkdir -p $O; rm -f $O/*
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0"
run() { name=$1; shift; env "$@" LIO_DEBUG_TIMING=1 timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
timeout 600 python -m pytest tests/test_gpu_contract.py -q -x 2>&1 | tail -3
run off LIO_RESIDENT_MOMENTS=0
run r2 LIO_RES_PER_LANE=2

run r4 LIO_RES_PER_LANE=4

for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 't_opt', d["stages_ms"]["t_opt"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
for n in off r2 r4; do echo "== $n"; grep "dogleg\|resident" $O/bench_$n.err | tail -2; done
