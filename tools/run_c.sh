O=gpurun_out/r3_c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python tools/stress_determinism.py 2000 > $O/stress.log 2>&1; tail -4 $O/stress.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_c/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['stages_ms']); print(json.dumps(d['roofline'])[:1500]); print(d['cpu_baseline']['value'] if d['cpu_baseline'] else None)
PY
