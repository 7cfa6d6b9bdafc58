"""End to end from t = 0 through the C-ABI only (PointProcessor -> PointOdometry -> /compact_data -> scan-to-map ->
window filling -> IMU initialisation -> sliding-window solves), CPU oracle.  Checks the stage machine of
Estimator::ProcessLaserOdom (Estimator.cc:430-618) and that the initialised estimator tracks the analytic trajectory."""
import numpy as np

from replay_util import run_from_zero, window_vs_truth


def test_oracle_runs_from_zero_and_tracks(oracle):
    W = 6
    rp, traj = run_from_zero(oracle, 26, W=W, Wo=3, init_window_factor=1, odom_io=2)
    events = [e["event"] for e in rp.log]
    # W frames fill the window, the (W+1)-th triggers the initialisation attempt
    assert events[:W] == ["filling"] * W
    assert "initialised" in events
    k0 = events.index("initialised")
    assert set(events[W:k0]) <= {"init_failed"}
    assert set(events[k0 + 1:]) == {"solved"} and len(events) - k0 >= 4
    st = rp.est.stage()
    assert st["inited"] and st["cir_buf_count"] == W
    # gravity aligned with -z of the world frame after RunInitialization (Estimator.cc:908-918)
    np.testing.assert_allclose(st["g_vec"], [0, 0, -9.805], atol=1e-9)
    np.testing.assert_allclose(st["R_WI"] @ st["R_WI"].T, np.eye(3), atol=1e-9)
    errs, w = window_vs_truth(rp, traj, W)
    assert errs[:, 0].max() < 0.08, errs     # m per 0.2 s step (about 1.2 m of motion)
    assert errs[:, 1].max() < 1.0, errs      # deg
    assert abs(np.linalg.norm(w["Vs"][W - 1]) - np.linalg.norm(traj.vel(rp.log[-1]["stamp"]))) < 0.5
    # the scan-to-scan odometry was switched to its packer mode at initialisation (A.18)
    assert not rp.odom_enabled


def test_init_window_factor_skips_frames(oracle):
    rp, _ = run_from_zero(oracle, 12, W=6, Wo=3, init_window_factor=2, odom_io=1)
    events = [e["event"] for e in rp.log]
    assert events[0::2] == ["skipped"] * len(events[0::2])   # laser_odom_recv_count_ % 2 != 0 (Estimator.cc:436-439)
    assert set(events[1::2]) == {"filling"}
    assert rp.est.stage()["cir_buf_count"] == len(events[1::2])
