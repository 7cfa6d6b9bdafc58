#!/bin/bash
# Round 4, GPU call 12: the opt-in one-launch VoxelGrid's tests after the box-mode fixes; the default bench with the threaded PointProcessor feed.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
(timeout 300 python -m pytest tests/test_gpu_vox_fused.py -q -x > $O/pytest_vox.log 2>&1; echo rc=$? >> $O/pytest_vox.log)
grep -E "passed|failed|rc=" $O/pytest_vox.log | tail -3
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err)
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("stages_ms"))
print(d["fed_gpu"]["point_processor"]["points"])
PY
tail -3 $O/bench.err
