"""ImuInitializer (ImuInitializer.cc:49-436): the oracle recovers the quantities the synthetic trajectory fixes
(gyro bias, gravity in the laser world frame, velocities, extrinsic rotation), and the product's host
implementation (csrc/host_init.h) agrees with it.  Host math only — no GPU needed."""
import math

import numpy as np
import pytest

from lio_amd import capi, synth


def _window(lib, n=16, frame_dt=0.2, rate=200.0, bg=(0.004, -0.003, 0.002), R_lb=None, t_lb=(0.05, -0.02, -0.08), t0=1.0, traj=None):
    traj = traj or synth.Trajectory()
    R_lb = np.eye(3) if R_lb is None else R_lb
    t_lb = np.asarray(t_lb, float)
    bg = np.asarray(bg, float)
    h = 1.0 / rate
    steps = int(round(frame_dt * rate))

    def laser_pose(t):
        R_wl = traj.rot(t) @ R_lb.T
        return R_wl, traj.pos(t) - R_wl @ t_lb

    R0, p0 = laser_pose(t0)
    transforms, pims, vels = [], [], []
    acc_prev, gyr_prev = traj.accel(t0), traj.gyro(t0) + bg
    for k in range(n):
        tk = t0 + k * frame_dt
        R, p = laser_pose(tk)
        transforms.append((synth.quat_from_rot(R0.T @ R), R0.T @ (p - p0)))
        vels.append(R0.T @ traj.vel(tk))
        pim = capi.Pim(lib, acc_prev, gyr_prev, np.zeros(3), np.zeros(3), g_norm=traj.g)
        if k > 0:
            for s in range(steps):
                t = tk - frame_dt + h * (s + 1)
                acc_prev, gyr_prev = traj.accel(t), traj.gyro(t) + bg
                pim.push_back(h, acc_prev, gyr_prev)
        pims.append(pim)
    g_l0 = R0.T @ np.array([0.0, 0.0, -traj.g])
    T_lb = (synth.quat_from_rot(R_lb), t_lb)
    return transforms, pims, T_lb, dict(vels=np.array(vels), g=g_l0, bg=bg)


def test_oracle_initialization_recovers_truth(oracle):
    tr, pims, T_lb, gt = _window(oracle)
    r = oracle.imu_initialization(tr, pims, T_lb)
    assert r["ok"]
    np.testing.assert_allclose(r["Bgs"], np.tile(gt["bg"], (len(tr), 1)), atol=2e-4)   # mid-point integration error only
    assert abs(np.linalg.norm(r["g"]) - 9.805) < 1e-9                                  # renormalised to g_norm (:212,:301)
    assert np.degrees(np.arccos(np.dot(r["g"], gt["g"]) / 9.805 ** 2)) < 0.5
    np.testing.assert_allclose(r["Vs"], gt["vels"], atol=0.08)
    # R_WI takes the inertial -z onto the gravity direction in the laser world frame (:303-312)
    np.testing.assert_allclose(r["R_WI"] @ np.array([0, 0, -1.0]), r["g"] / 9.805, atol=1e-12)
    np.testing.assert_allclose(r["R_WI"] @ r["R_WI"].T, np.eye(3), atol=1e-12)
    # the pims were re-propagated with the estimated gyro bias (:86-89)
    tr2, pims2, _, _ = _window(oracle)
    dq0, dq1 = pims2[3].get()["dq"], pims[3].get()["dq"]
    assert np.max(np.abs(dq0 - dq1)) > 1e-5


def test_oracle_rejects_short_or_wrong_windows(oracle):
    tr, pims, T_lb, _ = _window(oracle, n=5)   # window_size 4 < 5 (:99-102)
    assert not oracle.imu_initialization(tr, pims, T_lb)["ok"]
    # a grossly wrong gravity magnitude is rejected (:175): halve the specific force
    tr, pims, T_lb, _ = _window(oracle)
    slow = synth.Trajectory(g=4.0)
    tr_bad, pims_bad, T_lb, _ = _window(oracle, traj=slow)
    pims_mixed = [capi.Pim(oracle, slow.accel(1.0), slow.gyro(1.0), np.zeros(3), np.zeros(3), g_norm=9.805)] + pims_bad[1:]
    r = oracle.imu_initialization(tr_bad, pims_mixed, T_lb)
    assert not r["ok"]


def test_oracle_extrinsic_rotation(oracle):
    R_lb = synth.rot_zyx(0.4, -0.25, 0.3)
    wavy = synth.Trajectory(ang_scale=3.0)   # enough rotational excitation about all axes
    tr, pims, T_lb, _ = _window(oracle, R_lb=R_lb, bg=(0, 0, 0), traj=wavy)
    ok, q = oracle.imu_estimate_extrinsic_rotation(tr, pims, ([0, 0, 0, 1], T_lb[1]))
    assert ok
    R_est = synth.rot_from_quat(q)
    ang = np.degrees(np.arccos(np.clip((np.trace(R_est.T @ R_lb) - 1) / 2, -1, 1)))
    assert ang < 0.3, ang
    # yaw-only motion leaves the rotation about the yaw axis unobservable: rejected (:389-395)
    flat = synth.Trajectory(ang_scale=0.0)
    tr, pims, T_lb, _ = _window(oracle, R_lb=R_lb, bg=(0, 0, 0), traj=flat)
    ok, _ = oracle.imu_estimate_extrinsic_rotation(tr, pims, ([0, 0, 0, 1], T_lb[1]))
    assert not ok


@pytest.mark.parametrize("case", ["default", "rotated_extrinsic", "short"])
def test_product_host_initializer_matches_oracle(hip, oracle, case):
    kw = dict(default={}, rotated_extrinsic=dict(R_lb=synth.rot_zyx(0.1, 0.05, -0.2), traj=synth.Trajectory(ang_scale=2.0)), short=dict(n=7))[case]
    out = []
    for lib in (oracle, hip):
        tr, pims, T_lb, _ = _window(lib, **kw)
        r = lib.imu_initialization(tr, pims, T_lb, Bgs=np.full((len(tr), 3), 1e-4))
        ok, q = lib.imu_estimate_extrinsic_rotation(tr, pims, ([0, 0, 0, 1], T_lb[1]))
        out.append((r, ok, q, pims[2].get()))
    (ro, oko, qo, po), (rh, okh, qh, ph) = out
    assert ro["ok"] == rh["ok"] and oko == okh
    for key, tol in (("Bgs", 1e-10), ("g", 1e-7), ("Vs", 1e-7), ("R_WI", 1e-9)):
        np.testing.assert_allclose(rh[key], ro[key], rtol=0, atol=tol, err_msg=key)
    assert min(np.max(np.abs(qh - qo)), np.max(np.abs(qh + qo))) < 1e-6
    np.testing.assert_allclose(ph["dq"], po["dq"], atol=1e-12)
    np.testing.assert_allclose(ph["jac"], po["jac"], atol=1e-10)
