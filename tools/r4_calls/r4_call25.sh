#!/bin/bash
# Round 4, GPU call 25: KNN_BATCH 8 vs 4 (candidate loads in flight per lane of the search kernels), A/B builds of the library.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
run() { name=$1; shift; env "$@" timeout 200 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run b8 LIO_HIP_LIB=$R/lio-mapping_amd/csrc/ab/liblio_hip_b8.so
run b4 A=1
run b8b LIO_HIP_LIB=$R/lio-mapping_amd/csrc/ab/liblio_hip_b8.so
run b4b A=1
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
(LIO_HIP_LIB=$R/lio-mapping_amd/csrc/ab/liblio_hip_b8.so timeout 300 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py -q -x > $O/pytest_b8.log 2>&1; tail -2 $O/pytest_b8.log)
