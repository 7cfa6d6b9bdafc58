"""The oracle AND the product's host code against THE REFERENCE'S OWN factor sources.

tests/golden/ref_factor_vectors.npz holds seeded inputs and the outputs of hyye/lio-mapping's IntegrationBase.h, ImuFactor.h,
PivotPointPlaneFactor.cc, PriorFactor.cc and PoseLocalParameterization.cc, compiled where they lie against the stand-in
headers of oracle/ref_shim (`make -C oracle ref`; tests/golden/make_ref_factor_vectors.py; neither can run on the GPU box, the
vectors can).  This pins what the oracle otherwise only restates — the formulas, constants, signs and block indices of SURVEY.md
§8(a) a9-a11, a15, a16, a24, including the reference's -0.1667 in F(0, 12), the 0.5 factors of V, the PriorFactor's
Q^-1 * rot_ in the Jacobian and the hard-wired 1000 / 0.1 weights — down to the rounding of a different but equivalent order of
floating-point operations (the shim is not Eigen; tolerances are relative 1e-9 or tighter on well-scaled quantities).

Both libraries are loaded on the CPU: these entry points are host code in the product too."""
import os

import numpy as np
import pytest

from lio_amd import capi

V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_factor_vectors.npz"))


@pytest.fixture(scope="module", params=["oracle", "product"])
def lib(request, oracle):
    if request.param == "oracle":
        return oracle
    return capi.load_hip()      # raises when the product library is missing: no substitution


def _close(a, b, rtol, what):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, (what, err)
    return err


def _pim(lib, k):
    noise, dt, acc, gyr = V[f"pim{k}_noise"], V[f"pim{k}_dt"], V[f"pim{k}_acc"], V[f"pim{k}_gyr"]
    ba, bg = V[f"pim{k}_ba"], V[f"pim{k}_bg"]
    rep = int(V[f"pim{k}_repropagated"])
    # the intervals that were re-propagated were first integrated at OTHER biases; any start works, Repropagate resets everything
    ba0, bg0 = (ba + 0.01, bg - 0.001) if rep else (ba, bg)
    p = capi.Pim(lib, acc[0], gyr[0], ba0, bg0, *noise)
    for i in range(len(dt)):
        p.push_back(float(dt[i]), acc[i + 1], gyr[i + 1])
    if rep:
        p.repropagate(ba, bg)
    return p


@pytest.mark.parametrize("k", range(int(V["n_pim"])))
def test_preintegration_matches_the_reference(lib, k):
    """IntegrationBase::push_back / MidPointIntegration / Repropagate (IntegrationBase.h:104-312): delta_p, delta_q, delta_v,
    sum_dt, the 15 x 15 Jacobian and covariance after up to 200 samples."""
    g = _pim(lib, k).get()
    ref = V[f"pim{k}_state"]
    _close(g["dp"], ref[0:3], 1e-12, "delta_p")
    q = g["dq"] * np.sign(g["dq"][3] * ref[6])
    _close(q, ref[3:7], 1e-12, "delta_q")
    _close(g["dv"], ref[7:10], 1e-12, "delta_v")
    assert abs(g["sum_dt"] - ref[10]) < 1e-14
    _close(g["jac"], V[f"pim{k}_jac"], 1e-10, "jacobian")
    _close(g["cov"], V[f"pim{k}_cov"], 1e-10, "covariance")


@pytest.mark.parametrize("k", range(int(V["n_pim"])))
def test_imu_residual_and_factor_match_the_reference(lib, k):
    """IntegrationBase::Evaluate (:314-357) and ImuFactor::Evaluate (ImuFactor.h:53-168): raw residual, whitened residual
    (sqrt_info = LLT(covariance^-1).L^T) and the four whitened ambient Jacobians.  The whitening multiplies by the inverse of a
    covariance with entries down to 1e-12, so the whitened quantities carry the conditioning of that inverse (bound 1e-9).
    Measured in the build container: every quantity of this file is reproduced BIT FOR BIT by the oracle and by the product's host
    code (g++ -O3, hipcc -O3 and the reference at -O2, all without FMA contraction)."""
    p = _pim(lib, k)
    s = V[f"pim{k}_poses"]
    pose_i, sb_i, pose_j, sb_j = s[0:7], s[7:16], s[16:23], s[23:32]
    _close(p.evaluate(pose_i, sb_i, pose_j, sb_j), V[f"pim{k}_res_raw"], 1e-11, "raw residual")
    res, J = p.factor(pose_i, sb_i, pose_j, sb_j)
    _close(res, V[f"pim{k}_res"], 1e-9, "whitened residual")
    for q in range(4):
        _close(J[q], V[f"pim{k}_J{q}"], 1e-9, f"jacobian {q}")


def test_pivot_point_plane_factor_matches_the_reference(lib):
    """PivotPointPlaneFactor::Evaluate (PivotPointPlaneFactor.cc:43-137): residual and the three 1 x 7 Jacobians (pivot pose, pose
    i, extrinsic; the seventh column is zero)."""
    worst = 0.0
    for row_in, row_out in zip(V["ppp_in"], V["ppp_out"]):
        point, coeff, pose_p, pose_i, ex = row_in[0:3], row_in[3:7], row_in[7:14], row_in[14:21], row_in[21:28]
        res, (Jp, Ji, Jex) = lib.factor_ppp(point, coeff, pose_p, pose_i, ex)
        got = np.concatenate([[res], Jp, Ji, Jex])
        scale = max(np.abs(row_out).max(), 1.0)
        worst = max(worst, np.abs(got - row_out).max() / scale)
        assert Jp[6] == 0 and Ji[6] == 0 and Jex[6] == 0
    assert worst < 1e-12, worst


def test_prior_factor_and_pose_plus_match_the_reference(lib):
    """PriorFactor::Evaluate (PriorFactor.cc:35-67) with its 1000 / 0.1 weights and its LeftQuatMatrix(Q^-1 * rot_) block;
    PoseLocalParameterization::Plus (PoseLocalParameterization.cc:35-50)."""
    for row_in, row_out in zip(V["prior_in"], V["prior_out"]):
        res, J = lib.factor_prior(row_in[0:3], row_in[3:7], row_in[7:14])
        _close(res, row_out[:6], 1e-12, "prior residual")
        _close(J, row_out[6:], 1e-12, "prior jacobian")
    for row_in, row_out in zip(V["plus_in"], V["plus_out"]):
        out = lib.pose_plus(row_in[:7], row_in[7:])
        _close(out, row_out[:7], 1e-14, "pose plus")
        Jref = row_out[7:].reshape(7, 6)                       # ComputeJacobian: [I6; 0] — what the solver's tangent layout assumes
        np.testing.assert_array_equal(Jref, np.vstack([np.eye(6), np.zeros((1, 6))]))


def test_committed_vectors_are_what_the_reference_produces(tmp_path):
    """Build container only: rebuild oracle/_ref from /root/reference, regenerate the vectors and compare with the committed file
    (the vectors must not drift from the reference they claim to come from)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/src/factor"):
        pytest.skip("the reference tree is not on this machine")
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"], check=True)
    gen = os.path.join(root, "tests", "golden", "make_ref_factor_vectors.py")
    code = open(gen).read().replace('path = os.path.join(HERE, "ref_factor_vectors.npz")', f'path = {str(tmp_path / "v.npz")!r}')
    subprocess.run([sys.executable, "-c", code.replace("__file__", repr(gen))], check=True, capture_output=True)
    fresh = np.load(str(tmp_path / "v.npz"))
    assert sorted(fresh.files) == sorted(V.files)
    for k in V.files:
        np.testing.assert_array_equal(fresh[k], V[k], err_msg=k)
