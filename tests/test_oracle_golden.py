"""The oracle against everything the reference's own tests hold for this path:
  * the four angle identities asserted at test/test_point_processor/test_point_processor.cc:57-61;
  * the IMU fixture test/data/imu_pose_vel.txt (slice committed under tests/golden/, made by
    tests/golden/make_imu_fixture_slice.py): intent of test_imu_factor.cc:435-444 — the pre-integration
    residual evaluated at the fixture's ground truth is ~0.
Plus cross-checks of restated third-party semantics against numpy/scipy (parity with Eigen/PCL/Ceres
themselves is UNPINNED: they are absent from the reference tree and from this image)."""
import ctypes
import math
import os

import numpy as np

from lio_amd import capi, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_angle_identities(oracle):
    f = oracle.dll.orc_normalize_rad
    g = oracle.dll.orc_normalize_deg
    f.restype = g.restype = ctypes.c_double
    f.argtypes = g.argtypes = [ctypes.c_double]
    # EXPECT_DOUBLE_EQ = 4 ulp
    assert math.isclose(f(-3.4 - 2 * math.pi), -3.4 + 2 * math.pi, rel_tol=1e-15, abs_tol=1e-15)
    assert math.isclose(f(3.4 + 2 * math.pi), 3.4 - 2 * math.pi, rel_tol=1e-15, abs_tol=1e-15)
    assert math.isclose(g(-190 - 360), -190 + 360, rel_tol=1e-15)
    assert math.isclose(g(190 + 360), 190 - 360, rel_tol=1e-15)


def _fixture():
    return np.load(os.path.join(HERE, "golden", "imu_fixture_slice.npz"))["rows"]


def test_preintegration_on_reference_fixture(oracle):
    """INTERVAL = 20 samples per factor, g = 9.81, zero biases (test_imu_factor.cc:43-44,213,226)."""
    rows = _fixture()
    t, q, p, v, gyr, acc = rows[:, 0], rows[:, 1:5], rows[:, 5:8], rows[:, 8:11], rows[:, 11:14], rows[:, 14:17]
    pose = lambda k: np.concatenate([p[k], [q[k, 1], q[k, 2], q[k, 3], q[k, 0]]])
    sb = lambda k: np.concatenate([v[k], np.zeros(6)])
    worst = np.zeros(3)
    for k0 in range(0, 380, 20):
        pim = capi.Pim(oracle, acc[k0], gyr[k0], np.zeros(3), np.zeros(3), g_norm=9.81)
        for k in range(k0 + 1, k0 + 21):
            pim.push_back(t[k] - t[k - 1], acc[k], gyr[k])
        assert abs(pim.get()["sum_dt"] - 0.1) < 1e-5  # the fixture prints t with 6 significant digits
        r = pim.evaluate(pose(k0), sb(k0), pose(k0 + 20), sb(k0 + 20))
        worst = np.maximum(worst, [np.abs(r[0:3]).max(), np.abs(r[3:6]).max(), np.abs(r[6:9]).max()])
        assert np.all(r[9:] == 0)
    # mid-point integration of exact 200 Hz samples over 0.1 s: position 1e-4 m, rotation 1e-5 rad, velocity 2e-3 m/s
    assert worst[0] < 1e-4 and worst[1] < 1e-5 and worst[2] < 2e-3, worst


def test_preintegration_covariance_is_spd_and_growing(oracle):
    rows = _fixture()
    t, gyr, acc = rows[:, 0], rows[:, 11:14], rows[:, 14:17]
    pim = capi.Pim(oracle, acc[0], gyr[0], np.zeros(3), np.zeros(3), g_norm=9.81)
    tr = []
    for k in range(1, 41):
        pim.push_back(t[k] - t[k - 1], acc[k], gyr[k])
        c = pim.get()["cov"]
        np.testing.assert_allclose(c, c.T, atol=1e-18)
        tr.append(np.trace(c))
    assert np.all(np.diff(tr) > 0)
    assert np.linalg.eigvalsh(pim.get()["cov"]).min() > 0


def test_voxel_grid_semantics_vs_numpy(oracle):
    """SURVEY.md B.1 restated independently in numpy: one centroid per occupied voxel, ascending voxel index."""
    rng = np.random.default_rng(11)
    pts = rng.uniform(-5, 5, size=(3000, 4)).astype(np.float32)
    leaf = np.float32(0.4)
    inv = np.float32(1.0) / leaf
    out = oracle.voxel_grid(pts, float(leaf))
    ijk = np.floor(pts[:, :3] * inv).astype(np.int64)
    mn = np.floor(pts[:, :3].min(0) * inv).astype(np.int64)
    mx = np.floor(pts[:, :3].max(0) * inv).astype(np.int64)
    div = mx - mn + 1
    key = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * div[0] + (ijk[:, 2] - mn[2]) * div[0] * div[1]
    uk = np.unique(key)
    assert out.shape[0] == uk.shape[0]
    ref = np.stack([pts[key == k].astype(np.float64).mean(0) for k in uk])
    np.testing.assert_allclose(out, ref, atol=2e-5)


def test_knn_vs_scipy(oracle):
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(12)
    m = rng.uniform(-10, 10, size=(5000, 4)).astype(np.float32)
    q = rng.uniform(-10, 10, size=(300, 4)).astype(np.float32)
    idx, sqd = oracle.knn(m, q, 5)
    d, i = cKDTree(m[:, :3].astype(np.float64)).query(q[:, :3].astype(np.float64), k=5)
    np.testing.assert_array_equal(idx, i)  # no exact ties in random data
    np.testing.assert_allclose(sqd, d**2, rtol=1e-5)
    assert np.all(np.diff(sqd, axis=1) >= 0)  # ascending (B.2)


def test_plane_fit_vs_numpy_lstsq(oracle):
    """colPivHouseholderQr().solve on the 5x3 system == the least-squares solution (B.4)."""
    rng = np.random.default_rng(13)
    n_true = np.array([0.93, 0.2, -0.3])  # within the +-60 deg elevation FOV gate (Estimator.cc:1063-1086)
    n_true /= np.linalg.norm(n_true)
    base = rng.uniform(-0.4, 0.4, size=(400, 3))
    base -= np.outer(base @ n_true, n_true)
    plane_pts = (base + 4.0 * n_true + rng.normal(0, 0.003, base.shape)).astype(np.float32)
    m = np.zeros((400, 4), np.float32)
    m[:, :3] = plane_pts
    s = np.zeros((1, 4), np.float32)
    s[0, :3] = 4.0 * n_true + np.array([0.01, 0.02, 0.0])
    T = capi.TransformF.make([0, 0, 0, 1], [0, 0, 0])
    valid, coef, score = oracle.calculate_features(m, s, T)
    assert valid[0] == 1
    idx, _ = oracle.knn(m, s, 5)
    A = m[idx[0], :3].astype(np.float64)
    x = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
    nrm = np.linalg.norm(x)
    expect = np.concatenate([x / nrm, [1 / nrm]]) * score[0]
    np.testing.assert_allclose(coef[0], expect, atol=2e-4)


def test_marginalization_identities(oracle):
    """Intent of the commented check at MarginalizationFactor.cc:309-310: the prior's sqrt factor reproduces the
    Schur complement; here through the estimator: the prior is PSD and its gradient vanishes at x0 + (-H^+ b)."""
    from lio_amd import pipeline

    ds = synth.make_dataset("indoor", 6, 0.2, lidar=synth.Lidar(16, -15, 15, 600))
    clouds = [pipeline.feature_clouds(oracle, ds.lidar, f.scan) for f in ds.frames]
    cfg = pipeline.config_indoor(oracle, 4, 2)
    cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
    pipeline.set_extrinsic(cfg, ds)
    est = capi.Estimator(oracle, cfg)
    pipeline.init_window(est, oracle, ds, [c[0] for c in clouds], pos_sigma=0.005, rot_sigma=0.0005, vel_sigma=0.005)
    est.solve()   # first solve: IMU cost > 1e3 => turn_off => no prior yet (Estimator.cc:1938-1942,2040)
    est.slide()
    rep = pipeline.feed_frame(est, ds, 5, clouds[5][0], clouds[5][1])
    assert rep.marginalized == 1
    pr = est.prior()
    n = pr["n"]
    assert n == 6 * 2 + 15  # 6*Wo + 15
    w = np.linalg.eigvalsh(pr["JtJ"])
    assert w.min() > -1e-6 * w.max()
    assert pr["x0"].shape[0] == 7 + 9 + 7 + 7  # pose1, sb1, pose2, ex in ambient layout
