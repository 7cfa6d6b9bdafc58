#!/bin/bash
# Round 4, third GPU call: the one-launch VoxelGrid (k_vox_fused) — its own tests first, then the suite, then bench A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
(timeout 300 python -m pytest tests/test_gpu_vox_fused.py -q -s -x > $O/pytest_vox.log 2>&1; echo rc=$? >> $O/pytest_vox.log)
tail -15 $O/pytest_vox.log
if grep -q "rc=0" $O/pytest_vox.log; then
  (timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
  grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
fi
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0"
(LIO_VOX_FUSED=1 timeout 300 $B > $O/bench_fused1.json 2> $O/bench_fused1.err)
(LIO_VOX_FUSED=0 timeout 300 $B > $O/bench_fused0.json 2> $O/bench_fused0.err)
(LIO_VOX_FUSED=0 timeout 120 python - > $O/vox_legacy_timing.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "lio-mapping_amd"); sys.path.insert(0, "tests")
from lio_amd import capi
import test_gpu_vox_fused as t
hip = capi.load_hip()
t.test_one_launch_filter_timing(hip)
PY
)
cat $O/vox_legacy_timing.txt | tail -2
for f in $O/bench_fused1.json $O/bench_fused0.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -3 $O/bench_fused1.err
