#!/bin/bash
# Round 4, GPU call 9: A/B of the VoxelGrid sort (library merge path vs onesweep radix) and of the resident newest-frame rounds, non-debug benches.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4i; mkdir -p $O; cd $R
(LIO_VOX_SORT=onesweep timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k voxel > $O/pytest_vox_onesweep.log 2>&1; echo rc=$? >> $O/pytest_vox_onesweep.log)
grep -E "passed|failed|rc=" $O/pytest_vox_onesweep.log | tail -2
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run default A=1
run onesweep LIO_VOX_SORT=onesweep
run resrounds LIO_RESIDENT_ROUNDS=1
run default2 A=1
run onesweep2 LIO_VOX_SORT=onesweep
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
