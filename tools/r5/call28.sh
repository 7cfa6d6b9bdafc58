R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5final5
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log)
grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SECONDS=0
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err)
echo "bench.py default run: $SECONDS s"
python - "$O/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], 'stages', d.get("stages_ms"))
for p in d["batched"]["points"]:
    print("batched", p["windows"], p["value"], p["ms_per_batch_step"], {k.split(" ")[0]: v.get("device_ms") for k, v in p["stages"].items()})
PY
