O=gpurun_out/r3_g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_contract.py tests/test_gpu_parity.py tests/test_gpu_degenerate.py tests/test_gpu_end_to_end.py -q -x 2>&1 | tail -4
B="python bench.py --no-pmc --no-cpu-baseline --windows 4 --keyframes 0 --no-fed"
for m in 0 1; do LIO_RESIDENT_ROUNDS=$m LIO_DEBUG_TIMING=1 timeout 300 $B > $O/b$m.json 2> $O/b$m.err; python - $O/b$m.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["stages_ms"], 'batched', d["batched"])
PY
grep "resident rounds" $O/b$m.err | tail -2
done
