"""The oracle's marginalization against THE REFERENCE'S OWN MarginalizationInfo.

tests/golden/ref_marginalization_vectors.npz holds the priors that hyye/lio-mapping's src/factor/MarginalizationFactor.cc —
compiled where it lies, together with its ImuFactor.h and PivotPointPlaneFactor.cc, against the stand-ins of oracle/ref_shim
(`make -C oracle ref`) — produces when it is driven the way Estimator::SolveOptimization drives it (Estimator.cc:2152-2245) on the
three marginalizations of the sequence in tests/ref_marg_cases.py: ResidualBlockInfo::Evaluate with Ceres' Cauchy corrector, the
address-keyed block bookkeeping and drop sets, the four-thread A / b accumulation, the Schur complement through the
eigen-decomposition pseudo-inverse with its 1e-8 cut, the factorisation into linearized_jacobians / linearized_residuals,
GetParameterBlocks with the address shift, and — steps 2 and 3 — MarginalizationFactor::Evaluate on the previous prior (dx with the
quaternion sign rule).  The reference was fed the ORACLE's own post-solve states, features and previous prior, so the comparison
below is same-input: J^T J to 1e-8 of its largest entry, J^T r to 1e-6 (the 1e-8 eigenvalue cut sits inside the rounding noise of a
matrix with entries ~1e9: tests/golden/README.md), x0 exactly.  Stood in: Eigen's dense API and SelfAdjointEigenSolver (forwarded
to the oracle's Jacobi), Ceres' CauchyLoss."""
import os

import numpy as np
import pytest

from ref_marg_cases import run

V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_marginalization_vectors.npz"))


def test_oracle_marginalization_equals_the_reference(oracle):
    rows = run(oracle, lambda k, est, ds, prev: est.prior())
    assert len(rows) == int(V["steps"]) == 3
    for s, po in enumerate(rows):
        n, m = V[f"s{s}_nm"]
        assert po["n"] == n == 39 and m == 15
        JtJ, Jtr, x0 = V[f"s{s}_JtJ"], V[f"s{s}_Jtr"], V[f"s{s}_x0"]
        dj = np.abs(po["JtJ"] - JtJ).max() / np.abs(JtJ).max()
        dr = np.abs(po["Jtr"] - Jtr).max() / np.abs(Jtr).max()
        dx = np.abs(po["x0"] - x0).max()
        print(f"step {s}: |dJtJ| / max {dj:.2e}, |dJtr| / max {dr:.2e}, |dx0| {dx:.2e}")
        assert dj < 1e-8 and dr < 1e-6 and dx < 1e-12


def test_committed_vectors_are_what_the_reference_produces(tmp_path):
    """Build container only: rebuild oracle/_ref from /root/reference and regenerate (same-input: the generator replays the oracle)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/src/factor"):
        pytest.skip("the reference tree is not on this machine")
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"], check=True)
    gen = os.path.join(root, "tests", "golden", "make_ref_marginalization_vectors.py")
    out = str(tmp_path / "v.npz")
    code = open(gen).read().replace('path = os.path.join(HERE, "ref_marginalization_vectors.npz")', f"path = {out!r}").replace("__file__", repr(gen))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True)
    fresh = np.load(out)
    for k in V.files:
        np.testing.assert_allclose(fresh[k], V[k], rtol=1e-9, atol=1e-9 * max(1.0, float(np.abs(V[k]).max())), err_msg=k)
