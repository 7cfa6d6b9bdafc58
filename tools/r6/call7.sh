# round 6, call 7: the fused-rotation Jacobi of k_bw_marg_schur — tests, then the batched bench points
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_marg_device.py tests/test_gpu_batch.py tests/test_gpu_batch_scale.py -x -q 2>&1 | tail -4 > gpurun_out/r6/call7.log
python bench.py --no-cpu-baseline --no-pmc --no-fed --keyframes 0 --steps 20 > gpurun_out/r6/call7_bench.json 2>gpurun_out/r6/call7_bench.err
python - <<'PY' >> gpurun_out/r6/call7.log
import json
d = json.loads(open("gpurun_out/r6/call7_bench.json").read().strip().splitlines()[-1])
print("single window:", d["value"], d["unit"], "ms/step", d["ms_per_step"])
for p in d["batched"]["points"]:
    print(p["windows"], p.get("parity"), p.get("value"), p.get("ms_per_batch_step"), {k.split(" ")[0]: v.get("device_ms") for k, v in p.get("stages", {}).items()}, p.get("host_ms"))
PY
cat gpurun_out/r6/call7.log
