"""The device-only pieces of the device-resident dogleg (csrc/solve_step.h, launch B): the blocked L D L^T in LDS —
register-resident 16x16 diagonal block on one wave (v_readlane broadcasts), row-parallel triangular solves, fp64-MFMA
trailing updates, blocked back-substitution — against numpy / the oracle's Cholesky on random SPD systems at the sizes the
solver sees (D = 90 / 96 for opt window 5, 120 / 126 for 7, ragged sizes in between), and the whole solve with the host loop
(LIO_DEVICE_SOLVE=0) as the reference for the device loop."""
import numpy as np
import pytest

from lio_amd import capi, pipeline, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 15, 16, 17, 45, 90, 96, 102, 120, 126, 128])
def test_dense_spd_solve_matches_numpy(hip, oracle, n):
    rng = np.random.default_rng(n)
    J = rng.normal(size=(3 * n + 5, n)) * np.exp(rng.uniform(-3, 3, size=n))      # badly scaled columns, like the real H
    A = J.T @ J + 1e-9 * np.eye(n)
    b = rng.normal(size=n)
    x = hip.dense_spd_solve(A, b)
    xr = np.linalg.solve(A, b)
    xo = oracle.dense_spd_solve(A, b)
    scale = np.abs(xr).max()
    cond = np.linalg.cond(A)
    print(f"n={n} cond {cond:.1e} |x - numpy|/max {np.abs(x - xr).max() / scale:.1e}  oracle {np.abs(xo - xr).max() / scale:.1e}")
    # backward-stable: the residual is at rounding level whatever the conditioning
    assert np.abs(A @ x - b).max() <= 1e-9 * (np.abs(A).max() * np.abs(x).max() + np.abs(b).max())
    assert np.abs(x - xr).max() <= 1e-13 * cond * scale + 1e-12 * scale


def test_dense_spd_solve_rejects_indefinite(hip):
    A = np.eye(40)
    A[17, 17] = -1.0
    with pytest.raises(capi.LioError):
        hip.dense_spd_solve(A, np.ones(40))
    A = np.ones((33, 33))      # rank one: the second pivot is zero
    with pytest.raises(capi.LioError):
        hip.dense_spd_solve(A, np.ones(33))


def test_device_loop_equals_host_loop(hip, monkeypatch):
    """The same chain through the device-resident dogleg (LIO_DEVICE_SOLVE=1) and through the host loop: two estimators in
    lockstep, the device one handed the host one's states, extrinsic and prior before every step (tests/golden/README.md:
    a chain amplifies its inputs' differences).  Identical iteration counts, accepted / rejected steps, termination codes,
    convergence flags; cost traces within 1e-7; states within 1e-7 m after every step; priors equal to 1e-7."""
    from window_util import force_all, window_gap

    ds = synth.make_dataset("indoor", 12, 0.2)
    clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]

    def make(device):
        monkeypatch.setenv("LIO_DEVICE_SOLVE", "1" if device else "0")   # read when the estimator is created
        cfg = pipeline.config_indoor(hip, 4, 2)
        cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
        pipeline.set_extrinsic(cfg, ds)
        est = capi.Estimator(hip, cfg)
        pipeline.init_window(est, hip, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=3)
        return est

    ea, eb = make(True), make(False)

    def same(a, b):
        assert (a.iterations, a.successful_steps, a.termination) == (b.iterations, b.successful_steps, b.termination)
        assert (a.convergence_flag, a.turn_off, a.marginalized, a.n_lidar_residuals) == (b.convergence_flag, b.turn_off, b.marginalized, b.n_lidar_residuals)
        n = b.iterations + 1
        np.testing.assert_allclose(a.cost_trace[:n], b.cost_trace[:n], rtol=1e-7)
        np.testing.assert_allclose([a.cost_pim_before, a.cost_ppp_before, a.cost_marg_before], [b.cost_pim_before, b.cost_ppp_before, b.cost_marg_before],
                                   rtol=1e-9, atol=1e-12)

    same(ea.solve(), eb.solve())
    assert window_gap(ea.get_window(), eb.get_window())[0] < 1e-7
    ea.slide(); eb.slide()
    rejected = 0
    for k in range(ea.W + 1, 12):
        force_all(ea, eb, ds)
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1])
        rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        same(ra, rb)
        rejected += rb.iterations - rb.successful_steps
        g = window_gap(ea.get_window(), eb.get_window())
        assert g[0] < 1e-7 and g[1] < 1e-7, g
        pa, pb = ea.prior(), eb.prior()
        np.testing.assert_allclose(pa["JtJ"], pb["JtJ"], rtol=0, atol=1e-7 * np.abs(pb["JtJ"]).max())
    print(f"device loop == host loop over {12 - ea.W - 1} steps ({rejected} rejected trust-region steps on the way)")
