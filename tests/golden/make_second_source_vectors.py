"""Generates tests/golden/second_source_vectors.npz: inputs for the second-source checks of SURVEY.md Appendix B together with
what the CPU oracle returns on them TODAY (so the fixture also pins the oracle against later edits).

  python tests/golden/make_second_source_vectors.py

qr_*      5x3 plane-fit systems and 6x6 normal-equation systems (Estimator.cc:1027,1306), full rank and exactly rank deficient
marg_*    assembled (A, b) with a prescribed spectrum around the 1e-8 cut of MarginalizationFactor.cc:275-302
vox_*     clouds with points exactly on voxel faces, negative coordinates and non-finite points (B.1)
dl_*      the linearisations (J^T J, J^T r, cost) and per-iteration records of two oracle solves (one without, one with a
          marginalization prior) for the replay of the Ceres dogleg transliteration (B.3)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lio_amd import capi, pipeline, synth  # noqa: E402

orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)


def oracle_qr(A, b):
    A = np.ascontiguousarray(A, np.float32); b = np.ascontiguousarray(b, np.float32)
    x = np.zeros(A.shape[1], np.float32)
    assert orc.dll.orc_colpiv_qr_solve_f32(A.shape[0], A.shape[1], A.ctypes.data_as(fp), b.ctypes.data_as(fp), x.ctypes.data_as(fp)) == 0
    return x


def oracle_marg(A, b, m):
    A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64)
    n = A.shape[0] - m
    J, r = np.zeros((n, n)), np.zeros(n)
    assert orc.dll.orc_marginalize_schur(A.ctypes.data_as(dp), b.ctypes.data_as(dp), m, n, J.ctypes.data_as(dp), r.ctypes.data_as(dp)) == 0
    return J, r


def oracle_solve_dump(est):
    rep = capi.SolveReport()
    assert orc.dll.orc_est_solve_with_dump(C.c_void_p(est.h), C.byref(rep)) == 0
    n, nl, ni = C.c_int(), C.c_int(), C.c_int()
    orc.dll.orc_dump_sizes(C.byref(n), C.byref(nl), C.byref(ni))
    n, nl, ni = n.value, nl.value, ni.value
    H, g, cost = np.zeros((nl, n, n)), np.zeros((nl, n)), np.zeros(nl)
    for k in range(nl):
        c = C.c_double()
        orc.dll.orc_dump_lin(k, H[k].ctypes.data_as(dp), g[k].ctypes.data_as(dp), C.byref(c))
        cost[k] = c.value
    sc, fl, de = np.zeros((ni, 7)), np.zeros((ni, 3), np.int32), np.zeros((ni, n))
    for k in range(ni):
        orc.dll.orc_dump_it(k, sc[k].ctypes.data_as(dp), fl[k].ctypes.data_as(C.POINTER(C.c_int)), de[k].ctypes.data_as(dp))
    return dict(H=H, g=g, cost=cost, scalars=sc, flags=fl, delta=de, iterations=rep.iterations, termination=rep.termination,
                trace=np.array(rep.cost_trace[: rep.iterations + 1]))


def main():
    rng = np.random.default_rng(20260924)
    out = {}
    # ---- QR
    qrA, qrb = [], []
    for _ in range(12):    # plane fits: five neighbours of a plane, A n = -1
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        base = rng.normal(size=(5, 3)) * 0.4
        base -= np.outer(base @ nrm, nrm)
        qrA.append((base + rng.uniform(2, 30) * nrm + rng.normal(0, 0.01, (5, 3))).astype(np.float32)); qrb.append(-np.ones(5, np.float32))
    A = rng.normal(size=(5, 3)).astype(np.float32); A[:, 2] = A[:, 0]                     # duplicate column
    qrA.append(A); qrb.append(-np.ones(5, np.float32))
    A = rng.normal(size=(5, 3)).astype(np.float32); A[:, 1] = 0                           # zero column
    qrA.append(A); qrb.append(rng.normal(size=5).astype(np.float32))
    t = np.linspace(0, 1, 5)[:, None].astype(np.float32)                                    # collinear points: rank 2
    qrA.append((np.array([[1, 2, 3]], np.float32) + t * np.array([[0.5, -1, 0.25]], np.float32))); qrb.append(-np.ones(5, np.float32))
    out["qr53_A"], out["qr53_b"] = np.stack(qrA), np.stack(qrb)
    out["qr53_x_oracle"] = np.stack([oracle_qr(a, b) for a, b in zip(qrA, qrb)])
    qA, qb = [], []
    for k in range(10):    # 6x6 normal equations of a laser-odom round
        Jm = rng.normal(size=(400, 6)).astype(np.float32) * np.array([30, 30, 30, 1, 1, 1], np.float32)
        qA.append((Jm.T @ Jm).astype(np.float32)); qb.append((Jm.T @ rng.normal(size=400).astype(np.float32)).astype(np.float32))
    Jm = rng.normal(size=(50, 6)).astype(np.float32); Jm[:, 5] = Jm[:, 4]                # degenerate direction: two equal columns
    qA.append((Jm.T @ Jm).astype(np.float32)); qb.append((Jm.T @ rng.normal(size=50).astype(np.float32)).astype(np.float32))
    out["qr66_A"], out["qr66_b"] = np.stack(qA), np.stack(qb)
    out["qr66_x_oracle"] = np.stack([oracle_qr(a, b) for a, b in zip(qA, qb)])
    # ---- marginalization tail: prescribed spectra on both eigen-steps
    m, n = 15, 21
    for tag, spec_m, spec_s in (("a", np.geomspace(1e2, 1e6, 15), np.geomspace(1e-3, 1e5, 21)),
                                ("b", np.r_[np.geomspace(1.0, 1e5, 12), 1e-10, 1e-12, 0.0], np.r_[np.geomspace(1e-2, 1e4, 17), 3e-7, 2e-9, 1e-11, 0.0])):
        Qm, _ = np.linalg.qr(rng.normal(size=(m, m))); Qs, _ = np.linalg.qr(rng.normal(size=(n, n)))
        Amm = (Qm * spec_m) @ Qm.T
        S = (Qs * spec_s) @ Qs.T
        range_m = Qm[:, spec_m > 1e-8]
        Arm = rng.normal(size=(n, range_m.shape[1])) @ range_m.T * 3.0       # couplings inside the range of Amm (a Gram matrix always has them there)
        Amm_pinv = (Qm * np.where(spec_m > 1e-8, 1 / np.where(spec_m > 1e-8, spec_m, 1), 0)) @ Qm.T
        A = np.zeros((m + n, m + n)); A[:m, :m] = Amm; A[m:, :m] = Arm; A[:m, m:] = Arm.T; A[m:, m:] = S + Arm @ Amm_pinv @ Arm.T
        b = np.r_[range_m @ rng.normal(size=range_m.shape[1]), Qs[:, spec_s > 1e-8] @ rng.normal(size=int((spec_s > 1e-8).sum()))]
        J, r = oracle_marg(A, b, m)
        out[f"marg_{tag}_A"], out[f"marg_{tag}_b"], out[f"marg_{tag}_m"] = A, b, np.array(m)
        out[f"marg_{tag}_J_oracle"], out[f"marg_{tag}_r_oracle"] = J, r
    # ---- VoxelGrid: faces, negatives, non-finite
    leaf = 0.4
    k = rng.integers(-12, 12, size=(600, 3)).astype(np.float32)
    on_faces = (k * np.float32(leaf)).astype(np.float32)                                      # exactly representable multiples of the leaf, both signs
    inside = rng.uniform(-5, 5, size=(1500, 3)).astype(np.float32)
    pts = np.zeros((2100 + 40, 4), np.float32)
    pts[:600, :3], pts[600:2100, :3] = on_faces, inside
    pts[2100:, :3] = on_faces[:40] + np.float32(1e-7)                                          # one ulp-ish inside the face
    pts[:, 3] = rng.uniform(0, 64, size=len(pts)).astype(np.float32)
    pts[rng.integers(0, len(pts), 15), 0] = np.nan
    pts[rng.integers(0, len(pts), 5), 2] = np.inf
    out["vox_pts"], out["vox_leaf"] = pts, np.array(leaf)
    out["vox_out_oracle"] = orc.voxel_grid(pts, leaf)
    # ---- dogleg traces: a small window, first solve (no prior) and the solve after one marginalization
    ds = synth.make_dataset("indoor", 7, 0.2, lidar=synth.Lidar(16, -15, 15, 600))
    clouds = [pipeline.feature_clouds(orc, ds.lidar, f.scan) for f in ds.frames]
    cfg = pipeline.config_indoor(orc, 4, 2)
    cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
    pipeline.set_extrinsic(cfg, ds)
    for tag, sig in (("near", (0.005, 0.0005, 0.005)), ("far", (0.15, 0.02, 0.2))):
        est = capi.Estimator(orc, cfg)
        pipeline.init_window(est, orc, ds, [c[0] for c in clouds], pos_sigma=sig[0], rot_sigma=sig[1], vel_sigma=sig[2])
        d1 = oracle_solve_dump(est)
        est.slide()
        f = ds.frames[5]
        for j in range(f.imu_dt.shape[0]):
            est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
        est.push_frame(capi.TransformF.make([0, 0, 0, 1], [0, 0, 0]), clouds[5][0], clouds[5][1], f.t)
        d2 = oracle_solve_dump(est)
        for name, d in ((f"dl_{tag}1", d1), (f"dl_{tag}2", d2)):
            for key, v in d.items():
                out[f"{name}_{key}"] = np.asarray(v)
            print(name, "n", d["H"].shape[1], "linearisations", len(d["cost"]), "iterations", d["iterations"], "termination", d["termination"],
                  "accepted", int(d["flags"][:, 2].sum()), "of", len(d["flags"]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "second_source_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
