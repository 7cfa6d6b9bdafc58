"""Drives lio_amd.replay.Replay from t = 0 over synthetic motion-distorted sweeps + analytic IMU (shared by the CPU
end-to-end test of the oracle and the GPU parity test)."""
import numpy as np

from lio_amd import capi, pipeline, replay, synth


def run_from_zero(lib, n_sweeps, W=6, Wo=3, init_window_factor=1, odom_io=2, kind="indoor", imu_rate=200.0, sweeps=None, traj=None, t0=1.0,
                  configure=None, on_step=None, est_factory=None, tap=None):
    """traj / t0: the trajectory object behind `sweeps` when it is not the kind's default (e.g. synth.FixtureTrajectory with
    t0 = 0); configure(cfg) may edit the estimator config; on_step(rp, k, log_entry) is called after every processed message; est_factory(cfg) replaces
    the library's estimator by another object with the same methods (tests/ref_est_util.py: the reference's own Estimator); tap(cfg, kind, ...)
    sees every raw IMU / compact message in arrival order (tests/test_gpu_dropin.py feeds them to the drop-in class's ROS callbacks)."""
    if sweeps is None:
        sweeps = synth.make_sweeps(kind, n_sweeps)
    sw, pose_fn, lid = sweeps
    traj_in = traj
    if kind == "indoor":
        traj = synth.Trajectory()   # the indoor trajectory of make_sweeps
        cfg = pipeline.config_indoor(lib, W, Wo)
        cfg.transform_lb = capi.TransformF.make([0, 0, 0, 1], [0.0, 0.0, -0.081939])
    else:                           # HDL-64E, outdoor_test_config_64.yaml, the outdoor trajectory of make_sweeps
        import math

        traj = synth.Trajectory(rx=45.0, ry=60.0, rz=0.3, cx=15.0, cy=15.0, cz=2.2, Kz=2 * math.pi / 5.0, g=9.80, ang_scale=0.3)
        cfg = pipeline.config_outdoor64(lib, W, Wo)
        cfg.transform_lb = capi.TransformF.make([0, 0, 0, 1], [-8.086759e-01, 3.195559e-01, -7.997231e-01])
    traj = traj_in or traj
    cfg.init_window_factor = init_window_factor
    cfg.extrinsic_stage = 1
    if configure:
        configure(cfg)
    rp = replay.Replay(lib, cfg, lid, odom_io=odom_io, tap=(lambda *m: tap(cfg, *m)) if tap else None)
    if est_factory:
        rp.est = est_factory(cfg)
    h = 1.0 / imu_rate
    t_imu = t0
    for k, s in enumerate(sw[:n_sweeps]):
        t_end = t0 + 0.1 * (k + 1)
        while t_imu <= t_end + h + 1e-9:
            rp.add_imu(t_imu, traj.accel(t_imu), traj.gyro(t_imu))
            t_imu += h
        n0 = len(rp.log)
        rp.add_sweep(s, t_end)
        for e in rp.log[n0:]:
            e["window"] = rp.est.get_window() if e["inited"] else None
            if on_step:
                on_step(rp, k, e)
    return rp, traj


def window_vs_truth(rp, traj, W):
    """Relative body motion between consecutive window slots against the analytic trajectory.  After SlideWindow slot i
    holds the frame of the (W - i)-th last processed message and slot W duplicates slot W-1 (Estimator.cc:2646-2651)."""
    w = rp.est.get_window()
    stamps = [e["stamp"] for e in rp.log if e["event"] != "skipped"][-W:]
    errs = []
    for i in range(W - 1):
        Ri, pi = traj.rot(stamps[i]), traj.pos(stamps[i])
        Rj, pj = traj.rot(stamps[i + 1]), traj.pos(stamps[i + 1])
        g = Ri.T @ (pj - pi)
        e = w["Rs"][i].T @ (w["Ps"][i + 1] - w["Ps"][i])
        Re, Rg = w["Rs"][i].T @ w["Rs"][i + 1], Ri.T @ Rj
        ang = np.degrees(np.arccos(np.clip((np.trace(Re.T @ Rg) - 1) / 2, -1, 1)))
        errs.append((np.linalg.norm(e - g), ang))
    return np.array(errs), w
