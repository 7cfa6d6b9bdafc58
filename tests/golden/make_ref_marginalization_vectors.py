#!/usr/bin/env python
"""Generates tests/golden/ref_marginalization_vectors.npz: the priors THE REFERENCE'S OWN MarginalizationInfo produces
(src/factor/MarginalizationFactor.cc + ImuFactor.h + PivotPointPlaneFactor.cc compiled where they lie into
oracle/_ref/libref_factors.so, `make -C oracle ref`) on the sequence of tests/ref_marg_cases.py — ResidualBlockInfo::Evaluate with
Ceres' Cauchy corrector, the address-keyed block bookkeeping, drop sets, the four-thread A / b accumulation, the Schur complement
through the eigen-decomposition pseudo-inverse, the 1e-8 eigenvalue cut, GetParameterBlocks with the address shift, and — from the
second step on — MarginalizationFactor::Evaluate on the previous prior.  Stored per step, in the canonical kept order: J^T J, J^T r
(of linearized_jacobians / linearized_residuals), the kept blocks' x0, n and m.  Stood in: Eigen's dense API and
SelfAdjointEigenSolver (forwarded to the oracle's Jacobi), Ceres' CauchyLoss.  Runs only in the build container."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
from lio_amd import capi  # noqa: E402
from ref_marg_cases import WO, marg_inputs, run  # noqa: E402

ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_factors.so"))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
ref.ref_pim_create.restype = C.c_void_p
ref.ref_pim_create.argtypes = [dp] * 5
ref.ref_pim_destroy.argtypes = [C.c_void_p]
ref.ref_pim_push.argtypes = [C.c_void_p, C.c_double, dp, dp]
ref.ref_marginalize.restype = C.c_int
ref.ref_marginalize.argtypes = [C.c_int, dp, dp, dp, C.c_int, dp, dp, C.c_int, ip, ip, dp, C.c_void_p, ip, dp, dp, ip, dp, dp, ip, ip, ip, ip, ip, dp, C.c_int]


def P(a):
    return None if a is None else np.ascontiguousarray(a, np.float64).ctypes.data_as(dp)


def I(a):
    return None if a is None else a.ctypes.data_as(ip)


def reference_prior(k, est, ds, prev):
    x = marg_inputs(k, est, ds)
    z = np.zeros(3)
    keep = [np.ascontiguousarray(v, np.float64) for v in (x["imu_prev"].imu_acc[-1], x["imu_prev"].imu_gyr[-1], z, x["noise"])]
    h = ref.ref_pim_create(P(keep[0]), P(keep[1]), P(keep[2]), P(keep[2]), P(keep[3]))
    for j in range(len(x["imu"].imu_dt)):
        a, g = np.ascontiguousarray(x["imu"].imu_acc[j], np.float64), np.ascontiguousarray(x["imu"].imu_gyr[j], np.float64)
        ref.ref_pim_push(h, float(x["imu"].imu_dt[j]), P(a), P(g))
    feat_n = np.array([len(f[0]) for f in x["feats"]], np.int32)
    pts = np.ascontiguousarray(np.concatenate([f[0] for f in x["feats"]]), np.float64)
    cfs = np.ascontiguousarray(np.concatenate([f[1] for f in x["feats"]]), np.float64)
    poses, sbs, ex = (np.ascontiguousarray(v, np.float64) for v in (x["poses"], x["sbs"], x["ex"]))
    cap = 128
    m_out, nb = C.c_int(0), C.c_int(0)
    lin_jac, lin_res, x0 = np.zeros(cap * cap), np.zeros(cap), np.zeros(512)
    kind, index, off, size = (np.zeros(32, np.int32) for _ in range(4))
    if prev is None:
        n = ref.ref_marginalize(WO, P(poses), P(sbs), P(ex), 0, None, None, 0, None, None, None, h, I(feat_n), P(pts), P(cfs), C.byref(m_out), P(lin_jac),
                                P(lin_res), C.byref(nb), I(kind), I(index), I(off), I(size), P(x0), cap)
    else:
        pk = np.array([0, 1] + [0] * (WO - 1) + [2], np.int32)
        pi = np.array([0, 0] + list(range(1, WO)) + [0], np.int32)
        pj, pr, px = (np.ascontiguousarray(prev[key], np.float64) for key in ("lin_jac", "lin_res", "x0"))
        n = ref.ref_marginalize(WO, P(poses), P(sbs), P(ex), prev["n"], P(pj), P(pr), len(pk), I(pk), I(pi), P(px), h, I(feat_n), P(pts), P(cfs),
                                C.byref(m_out), P(lin_jac), P(lin_res), C.byref(nb), I(kind), I(index), I(off), I(size), P(x0), cap)
    ref.ref_pim_destroy(h)
    assert n > 0
    J, r = lin_jac[:n * n].reshape(n, n), lin_res[:n]
    JtJ, Jtr = J.T @ J, J.T @ r
    canon = [(0, 0), (1, 0)] + [(0, i) for i in range(1, WO)] + [(2, 0)]
    xoffs = np.concatenate([[0], np.cumsum(size[:nb.value])])
    perm, x0c = [], []
    for kk, ii in canon:
        b = [t for t in range(nb.value) if kind[t] == kk and (index[t] == ii or kk == 2)][0]
        ls = 6 if size[b] == 7 else size[b]
        perm += list(range(off[b], off[b] + ls))
        x0c.append(x0[xoffs[b]:xoffs[b] + size[b]])
    # the reference keeps its blocks in the iteration order of an unordered_map keyed by ADDRESS; the order below is what it was here
    return dict(n=n, m=m_out.value, JtJ=JtJ[np.ix_(perm, perm)], Jtr=Jtr[perm], x0=np.concatenate(x0c))


def main():
    oracle = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    rows = run(oracle, reference_prior)
    out = {"steps": np.array(len(rows))}
    for s, r in enumerate(rows):
        for key in ("JtJ", "Jtr", "x0"):
            out[f"s{s}_{key}"] = r[key]
        out[f"s{s}_nm"] = np.array([r["n"], r["m"]])
    path = os.path.join(HERE, "ref_marginalization_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", [(r["n"], r["m"]) for r in rows])


if __name__ == "__main__":
    main()
