"""The PRODUCT's host solver against the reference's own solve, on the CPU (no GPU, no oracle in between).

tests/golden/ref_solve_problems.npz holds four sliding-window problems exactly as hyye/lio-mapping's Estimator::SolveOptimization handed
them to ceres::Solve during a replay (dumped from the reference's own Estimator.cc compiled where it lies, see
tests/golden/make_ref_solve_problems.py): parameter blocks, the raw IMU samples behind every ImuFactor, ~13.8 k PivotPointPlaneFactor
points and planes, the marginalization prior that went in, the extrinsic-prior constants — and what came out: the parameters after the
solve, the cost after every iteration, the prior the marginalization produced.  tests/host/ref_solve_check.hip builds the product's
host_solver.h / host_factors.h (host code only), forms the lidar moments the GPU kernels would return by their defining sums, and runs
solve_dogleg and marginalize on the same problem.

What this pins directly against the reference's factor classes (ImuFactor, PivotPointPlaneFactor under CauchyLoss, MarginalizationFactor,
PriorFactor): the product's moment form of the lidar normal equations (H_i = L S L^T), its IMU / prior blocks, the assembly, and its
marginalization.  The minimizer on the reference's side is the stand-in of oracle/ref_shim/ceres/problem.h (Ceres itself is absent), so
the iteration sequence is compared between two restatements of Ceres 1.14's dogleg.

Two of the problems are the hard ones of a replay: a 5 / 2 window, first without any prior (the absolute pose is a gauge freedom: the
scaled normal matrix has its smallest eigenvalue at the 1e-8 regularisation), then with a prior and a free, nearly unobservable
extrinsic.  Measured: first linearisation (H, g) equal to 1e-14 / 6e-13 relative; the same 10 iterations; every cost of the trace
within 2e-9 / 2e-7 relative; positions after the solve within 2e-8 / 1.4e-7 m (relative positions 2.5e-9 / 2.8e-7), velocities 1e-8 /
1.4e-6, the free extrinsic 9e-6 m; the new prior's JtJ within 1e-9 / 4e-9 of its largest entry, Jtr 3e-10 / 8e-7, x0 equal.
The third is a well-posed one (a 6 / 3 window with the extrinsic PriorFactor, ~20 k plane factors): first linearisation 9e-15 / 5e-12,
trace 2e-11, positions 3e-9 m, the new prior's JtJ 1e-9.  The fourth is BASELINE.json's headline configuration (HDL-64E, window 15 / 5,
~46 k plane factors, 96 unknowns): first linearisation 4e-14 / 4e-12, trace 4e-11, positions 6e-8 m, JtJ 2e-10.
Bounds: 1e-6 m and rad on poses (the north star asks 1e-4 after the same iteration count), 1e-5 on velocities, 1e-4 on the extrinsic,
1e-6 relative on the trace, 1e-7 / 1e-5 on JtJ / Jtr."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "ref_solve_problems.npz"))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ref_solve") / "ref_solve_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-mavx2", "-Wno-unused-function",
                    "-I", os.path.join(ROOT, "lio-mapping_amd", "csrc"), os.path.join(ROOT, "tests", "host", "ref_solve_check.hip"), "-o", out], check=True)
    return out


PROBLEMS = [("indoor_iwf2", 1), ("indoor_iwf2", 2), ("indoor_prior_factor", 2), ("outdoor64_15_5", 2)]


def pack(case, step):
    k = "%s/s%d/" % (case, step)
    h = G[k + "header"]
    Wo, has_prior = int(h[0]), int(h[2])
    parts = [h, G[k + "initial"], G[k + "para"]]
    for i in range(Wo):
        if k + "imu%d_head" % i in G.files:
            smp = G[k + "imu%d_samples" % i]
            parts += [[float(len(smp))], G[k + "imu%d_head" % i], smp.reshape(-1)]
        else:
            parts.append([-1.0])
    for i in range(1, Wo + 1):
        pts, coef = G[k + "pts%d" % i], G[k + "coef%d" % i]
        parts += [[float(len(pts))], pts.reshape(-1), coef.reshape(-1)]
    if has_prior:
        bl = G[k + "prior_in_blocks"]                      # kind, index, column offset in the reference's own order, ambient size
        parts += [[float(G[k + "prior_in_jac"].shape[0]), float(len(bl))]]
        idx = 0                                            # (the stored Jacobian is already in canonical column order)
        for b in bl:                                       # -> kind, index, size, column offset
            parts.append([float(b[0]), float(b[1]), float(b[3]), float(idx)])
            idx += 6 if b[3] == 7 else int(b[3])
        parts += [G[k + "prior_in_x0"], G[k + "prior_in_jac"].reshape(-1), G[k + "prior_in_res"]]
    return np.concatenate([np.asarray(p, np.float64).reshape(-1) for p in parts])


@pytest.mark.parametrize("split", ["0", "1"])   # LIO_SPLIT_FACTOR: the full factorisation / the speed-bias rows factored ahead of the moments
@pytest.mark.parametrize("case,step", PROBLEMS)
def test_product_host_solver_on_the_reference_problem(exe, tmp_path, case, step, split):
    k = "%s/s%d/" % (case, step)
    path = str(tmp_path / "problem.f64")
    pack(case, step).tofile(path)
    r = subprocess.run([exe, path], capture_output=True, text=True, env=dict(os.environ, LIO_SPLIT_FACTOR=split, LIO_CHECK_DUMP_HG="1"))
    assert r.returncode == 0, r.stderr
    out = {ln.split()[0]: np.array(ln.split()[1:], float) for ln in r.stdout.strip().split("\n")}
    # the first linearisation: the product's normal equations (lidar part from the moments, H_i = L S L^T) against J^T J / J^T r summed
    # block by block from the reference's own factor classes
    H0, g0 = G[k + "H0"], G[k + "g0"]
    H = out["H"].reshape(H0.shape)
    sc = np.sqrt(np.outer(np.diag(H0), np.diag(H0)))
    dH, dg = float((np.abs(H - H0) / sc).max()), float(np.abs(out["g"] - g0).max() / np.abs(g0).max())
    assert dH <= 1e-12 and dg <= 1e-10, (dH, dg)
    it, succ, term = (int(v) for v in out["summary"][:3])
    want_it, want_term = (int(v) for v in G[k + "iterations"])
    assert (it, term) == (want_it, want_term)
    trace = G[k + "trace"][:want_it + 1]
    np.testing.assert_allclose(out["trace"][:want_it + 1], trace, rtol=1e-6)
    Wo = int(G[k + "header"][0])
    d = np.abs(out["params"] - G[k + "final"])
    frames = d[:16 * (Wo + 1)].reshape(Wo + 1, 16)
    gap = float(frames[:, :7].max())
    assert gap <= 1e-6, gap                                          # positions and quaternions
    assert frames[:, 7:10].max() <= 1e-5 and frames[:, 10:].max() <= 1e-6   # velocities; biases
    assert d[16 * (Wo + 1):].max() <= 1e-4                           # the extrinsic (constant in the first problem)
    # the marginalization, linearised where the reference linearised it
    n = int(out["prior"][0])
    JtJ, want = out["JtJ"].reshape(n, n), G[k + "JtJ"]
    assert JtJ.shape == want.shape
    rj = float(np.abs(JtJ - want).max() / np.abs(want).max())
    rr = float(np.abs(out["Jtr"] - G[k + "Jtr"]).max() / np.abs(G[k + "Jtr"]).max())
    assert rj <= 1e-7 and rr <= 1e-5, (rj, rr)
    np.testing.assert_allclose(out["x0"], G[k + "x0"], rtol=0, atol=1e-15)
    print(case, "step", step, "split", split, "first linearisation dH %.1e dg %.1e" % (dH, dg), "has prior", int(G[k + "header"][2]), "iterations", it, "param gap %.1e" % gap,
          "trace gap %.1e" % float(np.abs(out["trace"][:want_it + 1] / trace - 1).max()), "JtJ %.1e Jtr %.1e" % (rj, rr))
