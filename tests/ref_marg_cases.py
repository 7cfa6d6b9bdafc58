"""The solve / marginalize sequence on which the reference's own MarginalizationInfo was run for
tests/golden/ref_marginalization_vectors.npz — shared by the generator (tests/golden/make_ref_marginalization_vectors.py, build
container only) and tests/test_ref_marginalization_vectors.py.

One indoor VLP-16 window (8 / 4) is solved by the ORACLE estimator three times (push, solve, slide).  Before every marginalization
the generator hands the reference's MarginalizationInfo exactly what Estimator::SolveOptimization hands it (Estimator.cc:2152-2245):
the window's parameter blocks after the solve, the previous prior as a MarginalizationFactor, the IMU factor of the first interval,
one PivotPointPlaneFactor per lidar feature under CauchyLoss(1.0) — all read back from the oracle — and stores what comes out in the
canonical kept order (pose 0, speed-bias 0, pose 1 .. pose Wo-1, extrinsic)."""
import numpy as np

from lio_amd import capi
from window_util import make_pair

W, WO = 8, 4
STEPS = (W + 1, W + 2, W + 3)


def quat_from_R(R):
    """Eigen's Quaternion(Matrix3) (Shepperd branches), x y z w"""
    t = np.trace(R)
    if t > 0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        return np.array([(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t, w])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * t
    t = 0.5 / t
    q[3] = (R[k, j] - R[j, k]) * t
    q[j] = (R[j, i] + R[i, j]) * t
    q[k] = (R[k, i] + R[i, k]) * t
    return q


def run(lib, on_step):
    """Drives `lib`'s estimator through the sequence; on_step(k, est, ds, prev_prior_factor) is called after every solve that
    marginalized, before the slide.  Returns the list of what on_step returned."""
    ds, clouds, (est,) = make_pair((lib,), "indoor", W, WO, W + 4, 0.2)
    rep = est.solve()
    assert rep.marginalized == 0 and rep.turn_off == 1          # the perturbed window's IMU cost is above 1e3 (Estimator.cc:1935)
    est.slide()
    out = []
    for k in STEPS:
        f = ds.frames[k]
        for j in range(f.imu_dt.shape[0]):
            est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
        est.push_frame(capi.TransformF.make([0, 0, 0, 1], [0, 0, 0]), clouds[k][0], clouds[k][1], f.t)
        prev = est.prior_factor()
        rep = est.solve()
        assert rep.marginalized == 1
        out.append(on_step(k, est, ds, prev))
        est.slide()
    return out


def marg_inputs(k, est, ds):
    """what the reference's MarginalizationInfo is fed for the marginalization of step k (all read back from the estimator)"""
    pivot = W - WO
    win = est.get_window()
    poses, sbs = np.zeros((WO + 1, 7)), np.zeros((WO + 1, 9))
    for i in range(WO + 1):
        q = quat_from_R(win["Rs"][pivot + i])
        q /= np.linalg.norm(q)
        poses[i] = np.concatenate([win["Ps"][pivot + i], q])
        sbs[i] = np.concatenate([win["Vs"][pivot + i], win["Bas"][pivot + i], win["Bgs"][pivot + i]])
    ex = np.concatenate([win["t_lb"], win["q_lb"]]).astype(float)
    fi = k - W + pivot + 1                     # window slot pivot + 1 holds this dataset frame
    feats = [est.features(pivot + i) for i in range(1, WO + 1)]
    return dict(poses=poses, sbs=sbs, ex=ex, imu_prev=ds.frames[fi - 1], imu=ds.frames[fi], feats=feats,
                noise=np.array([est.cfg.acc_n, est.cfg.gyr_n, est.cfg.acc_w, est.cfg.gyr_w, est.cfg.g_norm]))
