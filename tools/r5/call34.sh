R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5final4
mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_dev_solver.py tests/test_gpu_marg_device.py -q -x > $O/pytest_batch.log 2>&1; echo rc=$? >> $O/pytest_batch.log)
grep -E "passed|failed|rc=" $O/pytest_batch.log | tail -2
(timeout 600 python bench.py > $O/bench.json 2> $O/bench.err)
python - "$O/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
for p in d["batched"]["points"]:
    print("batched", p["windows"], p["value"], p["ms_per_batch_step"], {k.split(" ")[0]: v.get("device_ms") for k, v in p["stages"].items()})
PY
