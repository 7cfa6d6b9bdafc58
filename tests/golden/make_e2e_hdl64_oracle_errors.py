"""Generates tests/golden/e2e_hdl64_oracle_errors.json: the CPU oracle replayed from t = 0 on the full-size HDL-64E
scenario (window 15 / 5, odom_io 3, 62 sweeps) and its per-step errors against the analytic trajectory.  The GPU test
test_full_size_hdl64_from_zero_tracks_truth compares the product's errors with these (the oracle needs ~60 s of CPU for
this scenario, too slow to repeat inside the GPU suite).  Run from the repo root: python tests/golden/make_e2e_hdl64_oracle_errors.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from lio_amd import capi  # noqa: E402
from replay_util import run_from_zero, window_vs_truth  # noqa: E402

lib = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
rp, traj = run_from_zero(lib, 62, W=15, Wo=5, init_window_factor=1, odom_io=3, kind="outdoor")
errs, w = window_vs_truth(rp, traj, 15)
out = {
    "events": [e["event"] for e in rp.log],
    "step_errors_m_deg": errs.tolist(),
    "n_lidar_residuals_last": int(rp.log[-1]["report"].n_lidar_residuals),
    "speed_last": float((w["Vs"][14] ** 2).sum() ** 0.5),
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "e2e_hdl64_oracle_errors.json"), "w"), indent=1)
print(out["events"][-8:], out["n_lidar_residuals_last"])
