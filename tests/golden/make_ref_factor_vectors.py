#!/usr/bin/env python
"""Generates tests/golden/ref_factor_vectors.npz: seeded random inputs and the outputs of THE REFERENCE'S OWN factor code on them.

oracle/_ref/libref_factors.so (`make -C oracle ref`, needs /root/reference) is the reference's IntegrationBase.h, ImuFactor.h,
PivotPointPlaneFactor.cc, PriorFactor.cc and PoseLocalParameterization.cc compiled where they lie against the stand-in headers
of oracle/ref_shim (a minimal dense-matrix / quaternion API in place of Eigen, Ceres' two base classes; see the header of
oracle/ref_shim/Eigen/Eigen for what that does and does not pin).  The vectors travel; the reference does not — so this script
runs only in the build container and its output is committed.  tests/test_ref_factor_vectors.py compares the oracle AND the
product's host code with them."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_factors.so"))
dp = C.POINTER(C.c_double)
ref.ref_pim_create.restype = C.c_void_p
ref.ref_pim_create.argtypes = [dp] * 5
ref.ref_pim_destroy.argtypes = [C.c_void_p]
ref.ref_pim_push.argtypes = [C.c_void_p, C.c_double, dp, dp]
ref.ref_pim_repropagate.argtypes = [C.c_void_p, dp, dp]
ref.ref_pim_get.argtypes = [C.c_void_p] + [dp] * 6
ref.ref_pim_evaluate.argtypes = [C.c_void_p] + [dp] * 5
ref.ref_imu_factor.argtypes = [C.c_void_p] + [dp] * 9
ref.ref_ppp_factor.argtypes = [dp] * 9
ref.ref_prior_factor.argtypes = [dp] * 5
ref.ref_pose_plus.argtypes = [dp] * 3
ref.ref_pose_jacobian.argtypes = [dp] * 2


def P(a):
    return a.ctypes.data_as(dp)


def rand_pose(rng, spread=2.0):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(scale=spread, size=3), q])          # [p, q_xyzw]


def near_pose(rng, pose, dp_=0.3, dr=0.05):
    d = rng.normal(scale=dr, size=3)
    dq = np.array([d[0] / 2, d[1] / 2, d[2] / 2, 1.0])
    x, y, z, w = pose[3:]
    a, b, c, s = dq
    q = np.array([w * a + x * s + y * c - z * b, w * b + y * s + z * a - x * c, w * c + z * s + x * b - y * a, w * s - x * a - y * b - z * c])
    return np.concatenate([pose[:3] + rng.normal(scale=dp_, size=3), q / np.linalg.norm(q)])


def main():
    rng = np.random.default_rng(20190406)
    out = {}
    # ---- pre-integration + ImuFactor: 6 intervals with different lengths, noises and biases
    n_pim = 6
    pim_in, pim_out = [], []
    for k in range(n_pim):
        noise = np.array([[0.1, 0.01, 0.0002, 2.0e-5, 9.805], [0.2, 0.02, 0.0002, 2.0e-5, 9.805], [0.05, 0.005, 0.001, 1e-4, 9.81]][k % 3])
        n = [20, 60, 40, 200, 8, 100][k]
        dt = np.full(n, [0.005, 0.005, 0.01, 0.0025, 0.005, 0.004][k]) + rng.uniform(-2e-4, 2e-4, n)
        acc = np.array([0.3, -0.2, 9.8]) + rng.normal(scale=0.5, size=(n + 1, 3))
        gyr = np.array([0.02, -0.01, 0.3]) + rng.normal(scale=0.1, size=(n + 1, 3))
        ba, bg = rng.normal(scale=0.02, size=3), rng.normal(scale=0.002, size=3)
        h = ref.ref_pim_create(P(acc[0].copy()), P(gyr[0].copy()), P(ba), P(bg), P(noise))
        for i in range(n):
            ref.ref_pim_push(h, float(dt[i]), P(acc[i + 1].copy()), P(gyr[i + 1].copy()))
        if k % 2 == 1:                                       # exercise Repropagate with new linearisation biases
            ba, bg = ba + rng.normal(scale=0.01, size=3), bg + rng.normal(scale=0.001, size=3)
            ref.ref_pim_repropagate(h, P(ba), P(bg))
        d_p, d_q, d_v, jac, cov, sdt = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(225), np.zeros(225), np.zeros(1)
        ref.ref_pim_get(h, P(d_p), P(d_q), P(d_v), P(jac), P(cov), P(sdt))
        # states: consistent with the pre-integrated motion up to a perturbation, biases away from the linearisation point
        pose_i = rand_pose(rng)
        sb_i = np.concatenate([rng.normal(scale=1.0, size=3), ba + rng.normal(scale=0.01, size=3), bg + rng.normal(scale=0.001, size=3)])
        pose_j = near_pose(rng, pose_i, dp_=1.0, dr=0.2)
        sb_j = np.concatenate([sb_i[:3] + rng.normal(scale=0.3, size=3), sb_i[3:6] + rng.normal(scale=1e-3, size=3), sb_i[6:] + rng.normal(scale=1e-4, size=3)])
        res_raw, res = np.zeros(15), np.zeros(15)
        J = [np.zeros(105), np.zeros(135), np.zeros(105), np.zeros(135)]
        ref.ref_pim_evaluate(h, P(pose_i), P(sb_i), P(pose_j), P(sb_j), P(res_raw))
        assert ref.ref_imu_factor(h, P(pose_i), P(sb_i), P(pose_j), P(sb_j), P(res), *[P(j) for j in J]) == 1
        ref.ref_pim_destroy(h)
        pim_in.append(dict(noise=noise, dt=dt, acc=acc, gyr=gyr, ba0=ba if k % 2 == 0 else None))
        out[f"pim{k}_noise"], out[f"pim{k}_dt"], out[f"pim{k}_acc"], out[f"pim{k}_gyr"] = noise, dt, acc, gyr
        out[f"pim{k}_ba"], out[f"pim{k}_bg"] = ba, bg                  # the FINAL linearisation biases (after Repropagate when k is odd)
        out[f"pim{k}_repropagated"] = np.array(k % 2)
        out[f"pim{k}_state"] = np.concatenate([d_p, d_q, d_v, sdt])
        out[f"pim{k}_jac"], out[f"pim{k}_cov"] = jac, cov
        out[f"pim{k}_poses"] = np.concatenate([pose_i, sb_i, pose_j, sb_j])
        out[f"pim{k}_res_raw"], out[f"pim{k}_res"] = res_raw, res
        for q in range(4):
            out[f"pim{k}_J{q}"] = J[q]
    out["n_pim"] = np.array(n_pim)
    # ---- PivotPointPlaneFactor: 64 (point, plane, pivot pose, pose i, extrinsic) tuples
    n = 64
    ppp_in, ppp_out = np.zeros((n, 3 + 4 + 21)), np.zeros((n, 1 + 21))
    for k in range(n):
        point = rng.normal(scale=10.0, size=3)
        w = rng.normal(size=3)
        w /= np.linalg.norm(w)
        coeff = np.concatenate([w * rng.uniform(0.5, 1.0), [rng.normal(scale=3.0)]])
        pose_p = rand_pose(rng, 5.0)
        pose_i = near_pose(rng, pose_p, dp_=2.0, dr=0.3)
        ex = near_pose(rng, np.array([0.0, 0.0, -0.1, 0, 0, 0, 1.0]), dp_=0.05, dr=0.05)
        res, Jp, Ji, Jex = np.zeros(1), np.zeros(7), np.zeros(7), np.zeros(7)
        assert ref.ref_ppp_factor(P(point), P(coeff), P(pose_p), P(pose_i), P(ex), P(res), P(Jp), P(Ji), P(Jex)) == 1
        ppp_in[k] = np.concatenate([point, coeff, pose_p, pose_i, ex])
        ppp_out[k] = np.concatenate([res, Jp, Ji, Jex])
    out["ppp_in"], out["ppp_out"] = ppp_in, ppp_out
    # ---- PriorFactor and PoseLocalParameterization
    n = 32
    pr_in, pr_out = np.zeros((n, 3 + 4 + 7)), np.zeros((n, 6 + 42))
    pl_in, pl_out = np.zeros((n, 7 + 6)), np.zeros((n, 7 + 42))
    for k in range(n):
        pose0 = rand_pose(rng)
        pose = near_pose(rng, pose0, dp_=0.1, dr=0.1)
        res, J = np.zeros(6), np.zeros(42)
        assert ref.ref_prior_factor(P(pose0[:3].copy()), P(pose0[3:].copy()), P(pose), P(res), P(J)) == 1
        pr_in[k], pr_out[k] = np.concatenate([pose0, pose]), np.concatenate([res, J])
        x, d = rand_pose(rng), rng.normal(scale=[0.5, 0.5, 0.5, 0.1, 0.1, 0.1])
        xo, Jp = np.zeros(7), np.zeros(42)
        ref.ref_pose_plus(P(x), P(d), P(xo))
        ref.ref_pose_jacobian(P(x), P(Jp))
        pl_in[k], pl_out[k] = np.concatenate([x, d]), np.concatenate([xo, Jp])
    out["prior_in"], out["prior_out"], out["plus_in"], out["plus_out"] = pr_in, pr_out, pl_in, pl_out
    path = os.path.join(HERE, "ref_factor_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
