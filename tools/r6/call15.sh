# round 6, call 15: full GPU suite + smoke + the driver's bench command
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r6/call15_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r6/call15_pytest.txt 2>&1
( time python bench.py > gpurun_out/r6/call15_bench.json 2> gpurun_out/r6/call15_bench.err ) 2>> gpurun_out/r6/call15_pytest.txt
cat gpurun_out/r6/call15_pytest.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/call15_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic")})
print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, d["cpu_baseline"].get("point_odometry_ms_per_scan") if d["cpu_baseline"] else None)
for p in d["batched"]["points"]:
    print(p["windows"], p.get("parity"), p.get("value"), p.get("ms_per_batch_step"), {k.split(" ")[0]: v.get("device_ms") for k, v in p.get("stages", {}).items()})
print("kf", d.get("keyframe_batch", {}).get("keyframes_per_s"), "ms_per_scan", d.get("ms_per_scan", {}).get("total"), d.get("ms_per_scan", {}).get("total_before_imu_init"))
PY
