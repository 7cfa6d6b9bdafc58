#!/bin/bash
# Round 4, GPU call 13: the bench with its steps looped inside the library, twice; the C-ABI hook against the Python loop.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4m; mkdir -p $O; cd $R
B="python bench.py --no-pmc --no-cpu-baseline --windows 0 --keyframes 0 --no-fed"
(timeout 200 $B > $O/bench_a.json 2> $O/bench_a.err)
(timeout 200 $B > $O/bench_b.json 2> $O/bench_b.err)
for f in $O/bench_a.json $O/bench_b.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d.get("stages_ms"))
PY
done
tail -2 $O/bench_a.err
(timeout 600 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py -q -x > $O/pytest_some.log 2>&1; tail -2 $O/pytest_some.log)
