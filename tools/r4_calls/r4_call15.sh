#!/bin/bash
# Round 4, GPU call 15: k_vox_fused with wave-aggregated counter updates, 32-cell groups, sparse prefixes, 64 M-cell table; phase stamps; suite; bench A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
(LIO_DEBUG_TIMING=1 timeout 300 python -m pytest tests/test_gpu_vox_fused.py -q -s -x > $O/pytest_vox.log 2>&1; echo rc=$? >> $O/pytest_vox.log)
grep -E "timing|passed|failed|us per filter|rc=" $O/pytest_vox.log | tail -12
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0"
(LIO_VOX_FUSED=1 timeout 300 $B > $O/bench_fused1.json 2> $O/bench_fused1.err)
(LIO_VOX_FUSED=0 timeout 300 $B > $O/bench_fused0.json 2> $O/bench_fused0.err)
(LIO_VOX_FUSED=1 LIO_DEBUG_TIMING=1 timeout 300 $B --steps 20 > $O/bench_fused1_dbg.json 2> $O/bench_fused1_dbg.err)
grep "k_vox_fused" $O/bench_fused1_dbg.err | tail -3
for f in $O/bench_fused1.json $O/bench_fused0.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
true
if false; then
  (timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
  grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
fi
