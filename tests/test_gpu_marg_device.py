"""Marginalization's dense tail on the device (csrc/marg_kernels.hip; MarginalizationFactor.cc:271-302): the Jacobi
eigensolver + fp64-MFMA Schur complement against the numpy second source (tests/golden/second_source.py), the oracle, and —
through the estimator with LIO_DEVICE_MARG=1 — against the host path and the oracle on a chain of solves."""
import os
import sys

import numpy as np
import pytest

from lio_amd import capi, pipeline, synth

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import second_source as ss  # noqa: E402

pytestmark = pytest.mark.gpu
V = np.load(os.path.join(os.path.dirname(__file__), "golden", "second_source_vectors.npz"))


def _graded_system(rng, m, n, n_null):
    """A = J^T J over m + n parameters whose Schur complement has eigenvalues from 1e-4 to ~1e4 and `n_null` exact null
    directions (the gauge freedom of a window): far enough from the absolute 1e-8 cut for the invariants to be pinned."""
    N = m + n
    rows = 4 * N
    J = rng.normal(size=(rows, N)) * 10.0 ** rng.uniform(-2, 1.5, size=N)[None, :]
    if n_null:
        # make the last n_null kept parameters exact copies of linear combinations of the others: null directions of A
        C = rng.normal(size=(N - n_null, n_null)) * 0.1
        J[:, N - n_null:] = J[:, : N - n_null] @ C
    A = J.T @ J
    b = J.T @ rng.normal(size=rows)
    return A, b


def _compare(J, r, s, J2, r2, s2, tol_g, tol_s=1e-7):
    kept, kept2 = int((s > 1e-8).sum()), int((s2 > 1e-8).sum())
    assert kept == kept2, (kept, kept2, np.sort(s)[:6], np.sort(s2)[:6])
    scale = np.abs(J2.T @ J2).max()
    np.testing.assert_allclose(J.T @ J, J2.T @ J2, rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(J.T @ r, J2.T @ r2, rtol=0, atol=tol_g * np.abs(J2.T @ r2).max())
    np.testing.assert_allclose(r @ r, r2 @ r2, rtol=max(tol_g, 1e-8))
    big = np.sort(s2)[-kept:]
    np.testing.assert_allclose(np.sort(s)[-kept:], big, rtol=tol_s)


@pytest.mark.parametrize("m,n,n_null", [(15, 21, 0), (15, 45, 0), (15, 45, 4), (15, 57, 4), (15, 80, 3), (6, 30, 2), (15, 1, 0)])
def test_device_schur_and_eigen_vs_numpy(hip, oracle, m, n, n_null):
    rng = np.random.default_rng(100 * m + n)
    A, b = _graded_system(rng, m, n, n_null)
    J, r, s = hip.marginalize_schur(A, b, m)
    assert np.all(np.diff(s) >= 0)                                   # ascending, like SelfAdjointEigenSolver
    J2, r2, s2 = ss.marginalize_schur(A, b, m)
    # the null directions come out as rounding noise (|s| ~ 1e-12 here) on both sides and are cut
    _compare(J, r, s, J2, r2, s2, 1e-7)
    Jo, ro, so = oracle.marginalize_schur(A, b, m)
    _compare(J, r, s, Jo, ro, np.where(so > 0, so, 0.0), 1e-7)


def test_device_schur_on_the_golden_vectors(hip):
    for tag in ("a", "b"):
        A, b, m = V[f"marg_{tag}_A"], V[f"marg_{tag}_b"], int(V[f"marg_{tag}_m"])
        J, r, s = hip.marginalize_schur(A, b, m)
        J2, r2, s2 = ss.marginalize_schur(A, b, m)
        # case b keeps an eigenvalue of 3e-7 that S's absolute rounding noise (~1e-11) moves by 1e-4 relative (test_second_source.py)
        _compare(J, r, s, J2, r2, s2, 1e-9 if tag == "a" else 2e-3, 1e-7 if tag == "a" else 1e-3)


def test_out_of_range_shapes_are_rejected(hip):
    A = np.eye(100)
    with pytest.raises(capi.LioError):
        hip.marginalize_schur(A, np.zeros(100), 15)                     # n = 85 > 80: the estimator keeps such windows on the host
    with pytest.raises(capi.LioError):
        hip.marginalize_schur(np.eye(40), np.zeros(40), 16)


@pytest.mark.parametrize("kind,W,Wo", [("indoor", 6, 3), ("outdoor", 15, 5)])
def test_estimator_with_device_marginalization(hip, oracle, monkeypatch, kind, W, Wo):
    """LIO_DEVICE_MARG=1: the prior of every step comes from the device kernel.  Same decisions and windows as the oracle over
    a chain of solves (teacher-forced, tests/window_util.py), and the priors agree on the order-equivariant invariants."""
    from window_util import assert_windows_close, force_all, make_pair
    monkeypatch.setenv("LIO_DEVICE_MARG", "1")
    ds, clouds, (ea, eb) = make_pair((hip, oracle), kind, W, Wo, W + 5, 0.2 if kind == "indoor" else 0.3)
    monkeypatch.delenv("LIO_DEVICE_MARG")
    for est in (ea, eb):
        est.solve()
        est.slide()
    force_all(ea, eb, ds)
    for k in range(W + 1, W + 5):
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1])
        rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        assert ra.marginalized == rb.marginalized == 1
        assert ra.iterations == rb.iterations and ra.termination == rb.termination
        assert_windows_close(ea.get_window(), eb.get_window())
        pa, pb = ea.prior(), eb.prior()
        assert pa["n"] == pb["n"] == 15 + 6 * Wo
        scale = np.abs(pb["JtJ"]).max()
        assert np.max(np.abs(pa["JtJ"] - pb["JtJ"])) / scale < 1e-4
        force_all(ea, eb, ds)
