"""The product's host linear algebra (Cholesky with its AVX-512 path, triangular solves, the symmetric eigensolver used by
the marginalization) against numpy.  The header is plain C++ (no HIP), so this runs on the CPU box: the dogleg step and the
prior are otherwise only exercised by the GPU parity tests."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hlinalg") / "hlinalg_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mavx2", "-I", os.path.join(ROOT, "lio-mapping_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "hlinalg_check.cc"), "-o", exe], check=True)
    return exe


def _run(exe, problems):
    text = []
    for H, g in problems:
        n = H.shape[0]
        text.append(str(n))
        text.append(" ".join(repr(float(v)) for v in H.ravel()))
        text.append(" ".join(repr(float(v)) for v in g))
    out = subprocess.run([exe], input="\n".join(text) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    res = []
    for k, (H, g) in enumerate(problems):
        n = H.shape[0]
        flags = [int(v) for v in out[7 * k].split()]
        rows = [np.array([float(v) for v in out[7 * k + r].split()]) for r in range(1, 7)]
        res.append((flags, rows[0], rows[1], rows[2], rows[3].reshape(n, n), rows[4], rows[5]))
    return res


def test_cholesky_solve_and_eigen_against_numpy(checker):
    rng = np.random.default_rng(11)
    problems = []
    for n in (1, 2, 3, 6, 7, 8, 9, 15, 45, 60, 96, 97, 126, 246):       # D = 96 / 126 / 246 are the solve sizes
        B = rng.normal(size=(n, n))
        problems.append((B @ B.T + n * np.eye(n), rng.normal(size=n)))
    res = _run(checker, problems)
    for (H, g), (flags, x, xp, w, V, xs, quad) in zip(problems, res):
        n = H.shape[0]
        assert flags == [1, 1, 1, 1]
        # chol_upper_from: out of place, H + diag(shift), H untouched (flag 4)
        ref_s = np.linalg.solve(H + np.diag(0.25 * np.diag(H) + 1e-3), g)
        assert np.abs(xs - ref_s).max() < 1e-11 * np.abs(ref_s).max() * n
        # sym_quad: x^T H x and H x, dispatched and portable
        Hg = H @ g
        np.testing.assert_allclose(quad[:2], g @ Hg, rtol=1e-12)
        np.testing.assert_allclose(quad[2::2], Hg, atol=1e-12 * np.abs(Hg).max())
        np.testing.assert_allclose(quad[3::2], Hg, atol=1e-12 * np.abs(Hg).max())
        ref = np.linalg.solve(H, g)
        scale = np.abs(ref).max()
        assert np.abs(x - ref).max() < 1e-11 * scale * n          # dispatched path (AVX-512 where the host has it)
        assert np.abs(xp - ref).max() < 1e-11 * scale * n         # portable path
        wr = np.linalg.eigvalsh(H)
        assert np.all(np.diff(w) >= 0)                            # ascending, like Eigen's SelfAdjointEigenSolver
        np.testing.assert_allclose(w, wr, rtol=1e-11, atol=1e-11 * abs(wr).max())
        np.testing.assert_allclose(V @ np.diag(w) @ V.T, H, atol=1e-10 * abs(H).max())   # eigenvectors are the COLUMNS
        np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-11)


def test_rank_deficient_and_indefinite_inputs(checker):
    rng = np.random.default_rng(5)
    B = rng.normal(size=(12, 5))
    semi = B @ B.T                                                 # rank 5: the gauge-deficient prior the marginalization sees
    indef = np.diag([1.0, -2.0, 3.0])
    res = _run(checker, [(semi, np.ones(12)), (indef, np.ones(3))])
    (f1, _, _, w1, V1, _, _), (f2, _, _, w2, _, _, _) = res
    assert f1[2] == 1 and np.sum(np.abs(w1) < 1e-9 * abs(w1).max()) == 7       # 7 (near-)zero eigenvalues, thresholded at 1e-8 by the caller
    np.testing.assert_allclose(V1 @ np.diag(w1) @ V1.T, semi, atol=1e-10 * abs(semi).max())
    assert f2[0] == 0 and f2[1] == 0                                            # not positive definite: the dogleg raises mu
    np.testing.assert_allclose(w2, [-2.0, 1.0, 3.0], atol=1e-12)


def test_lidar_linear_maps_reproduce_the_factor(tmp_path):
    """the moments kernel contracts z z^T; the host maps it to the 18x18 block with L.  L is read off four
    probe points of the factor's own Jacobian expressions — it must agree with the factor everywhere."""
    import shutil

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "linear_maps_check")
    subprocess.run([hipcc, "-O2", "-std=c++17", "-ffp-contract=off", "-mavx2", "--offload-arch=gfx950", "-I",
                    os.path.join(ROOT, "lio-mapping_amd", "csrc"), os.path.join(ROOT, "tests", "host", "linear_maps_check.hip"), "-o", exe],
                   check=True)
    jac_err, res_err = (float(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split())
    assert jac_err < 1e-11 and res_err < 1e-11
