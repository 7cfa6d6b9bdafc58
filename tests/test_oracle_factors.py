"""Finite-difference Jacobian checks of the oracle's factors — the reference's own (disabled) Check()
methods made into assertions: eps = 1e-6 and the q (x) DeltaQ(delta) perturbation convention of
src/factor/PivotPointPlaneFactor.cc:188-235 / PriorFactor.cc:69-118."""
import numpy as np
import pytest

from lio_amd import capi, synth


def rand_pose(rng, scale=1.0):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(size=3) * scale, q])


def num_jac(f, lib, blocks, eps=1e-6):
    """blocks: list of (array, kind) kind in {'pose','vec'}; returns list of d r / d local."""
    r0 = np.atleast_1d(f(*[b for b, _ in blocks]))
    out = []
    for bi, (b, kind) in enumerate(blocks):
        nloc = 6 if kind == "pose" else b.shape[0]
        J = np.zeros((r0.shape[0], nloc))
        for k in range(nloc):
            d = np.zeros(nloc)
            d[k] = eps
            if kind == "pose":
                bp = lib.pose_plus(b, d)
                # pose_plus normalises; the reference's Check() uses the unnormalised product — the
                # difference is O(eps^2)
            else:
                bp = b + d
            args = [x for x, _ in blocks]
            args[bi] = bp
            J[:, k] = (np.atleast_1d(f(*args)) - r0) / eps
        out.append(J)
    return out


def test_pivot_point_plane_jacobian(oracle):
    rng = np.random.default_rng(0)
    for _ in range(5):
        pp, pi, pex = rand_pose(rng, 3), rand_pose(rng, 3), rand_pose(rng, 0.3)
        point, coeff = rng.normal(size=3) * 5, rng.normal(size=4)
        r, js = oracle.factor_ppp(point, coeff, pp, pi, pex)
        nj = num_jac(lambda a, b, c: oracle.factor_ppp(point, coeff, a, b, c, jac=False)[0], oracle, [(pp, "pose"), (pi, "pose"), (pex, "pose")])
        for a, n in zip(js, nj):
            assert a[6] == 0.0
            np.testing.assert_allclose(a[:6], n[0], atol=2e-4, rtol=1e-4)


def test_prior_factor_jacobian(oracle):
    rng = np.random.default_rng(1)
    for _ in range(5):
        pose0 = rand_pose(rng)
        pose = oracle.pose_plus(pose0, rng.normal(size=6) * 0.05)
        r, J = oracle.factor_prior(pose0[:3], pose0[3:], pose)
        nj = num_jac(lambda a: oracle.factor_prior(pose0[:3], pose0[3:], a, jac=False)[0], oracle, [(pose, "pose")])
        # The reference writes LeftQuatMatrix(Q.inverse() * rot_) (PriorFactor.cc:56) where the exact
        # derivative is LeftQuatMatrix(rot_.inverse() * Q): same diagonal, skew part negated.  The oracle
        # restates the reference, so the rotation block is the TRANSPOSE of the numeric one.
        np.testing.assert_allclose(J[:3, :3], nj[0][:3, :3], atol=2e-3, rtol=1e-3)
        np.testing.assert_allclose(J[3:, 3:6].T, nj[0][3:, 3:6], atol=2e-4, rtol=1e-3)
        assert np.all(J[:3, 3:6] == 0) and np.all(J[3:, :3] == 0)
        assert np.all(J[:, 6] == 0)


def _pim_between(lib, traj, t0, t1, rate=200.0, g=9.805):
    h = 1.0 / rate
    n = int(round((t1 - t0) * rate))
    p = capi.Pim(lib, traj.accel(t0), traj.gyro(t0), np.zeros(3), np.zeros(3), acc_n=0.2, gyr_n=0.02, g_norm=g)
    for k in range(n):
        t = t0 + (k + 1) * h
        p.push_back(h, traj.accel(t), traj.gyro(t))
    return p


def test_preintegration_residual_at_ground_truth(oracle):
    """Intent of test_imu_factor.cc:435-444 on an analytic trajectory: residual at GT ~ 0."""
    traj = synth.Trajectory()
    t0, t1 = 2.0, 2.1
    p = _pim_between(oracle, traj, t0, t1)
    pose = lambda t: np.concatenate([traj.pos(t), synth.quat_from_rot(traj.rot(t))])
    sb = lambda t: np.concatenate([traj.vel(t), np.zeros(6)])
    r = p.evaluate(pose(t0), sb(t0), pose(t1), sb(t1))
    assert np.max(np.abs(r[:3])) < 1e-4 and np.max(np.abs(r[3:6])) < 1e-5 and np.max(np.abs(r[6:9])) < 1e-3
    assert np.all(r[9:] == 0)


def test_imu_factor_jacobian(oracle):
    rng = np.random.default_rng(2)
    traj = synth.Trajectory()
    t0, t1 = 3.0, 3.1
    p = _pim_between(oracle, traj, t0, t1)
    pose = lambda t: np.concatenate([traj.pos(t), synth.quat_from_rot(traj.rot(t))])
    pi = oracle.pose_plus(pose(t0), rng.normal(size=6) * 0.01)
    pj = oracle.pose_plus(pose(t1), rng.normal(size=6) * 0.01)
    sbi = np.concatenate([traj.vel(t0), rng.normal(size=6) * 0.01])
    sbj = np.concatenate([traj.vel(t1), rng.normal(size=6) * 0.01])
    r, js = p.factor(pi, sbi, pj, sbj)
    nj = num_jac(lambda a, b, c, d: p.factor(a, b, c, d, jac=False)[0], oracle, [(pi, "pose"), (sbi, "vec"), (pj, "pose"), (sbj, "vec")], eps=1e-7)
    for a, n, kind in zip(js, nj, ["pose", "vec", "pose", "vec"]):
        cols = 6 if kind == "pose" else 9
        scale = np.maximum(np.abs(n).max(), 1.0)
        np.testing.assert_allclose(a[:, :cols] / scale, n / scale, atol=2e-3)
        if kind == "pose":
            assert np.all(a[:, 6] == 0)


def test_imu_factor_whitening(oracle):
    """sqrt_info^T sqrt_info = cov^-1  =>  |r_white|^2 = r^T cov^-1 r  (ImuFactor.h:74-77)."""
    traj = synth.Trajectory()
    p = _pim_between(oracle, traj, 1.0, 1.2)
    pose = lambda t: np.concatenate([traj.pos(t), synth.quat_from_rot(traj.rot(t))])
    sb = lambda t: np.concatenate([traj.vel(t) + 0.01, np.full(6, 1e-3)])
    raw = p.evaluate(pose(1.0), sb(1.0), pose(1.2), sb(1.2))
    white, _ = p.factor(pose(1.0), sb(1.0), pose(1.2), sb(1.2), jac=False)
    cov = p.get()["cov"]
    np.testing.assert_allclose(white @ white, raw @ np.linalg.solve(cov, raw), rtol=1e-6)
