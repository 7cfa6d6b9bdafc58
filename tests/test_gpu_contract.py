"""GPU parity on the contract configurations SURVEY.md §8(d) names beyond the headline window (BASELINE.json `configs`):

  configs[2]  VLP-16 + 200 Hz IMU, window_size 15 / opt_window_size 5 (the compiled default, Estimator.h:78-79);
  indoor_test_config.yaml:12-13,68  window 12 / opt window 7 with keep_features = 1 through solve + slide + marginalization;
  configs[3] stress  HDL-64E with opt_window_size = window_size = 15 (tangent dimension D = 246);
  per-iteration cost trace within 1e-6 relative (Estimator.cc:1990-2021);
  a teacher-forced chain of 20 consecutive ProcessLaserOdom calls at 1e-4 m / 1e-4 rad.

The product runs through the C-ABI of liblio_hip.so; the oracle is the checker.

Chains are teacher-forced: before every step the product receives the oracle's window states, extrinsic and
marginalization prior (window_util.force_all), so each step is a comparison on the same inputs.  Without that a chain
measures the conditioning of the algorithm, not the implementation: the oracle run against ITSELF with a 1e-8 m
perturbation of the states differs by up to 4e-3 m after six HDL-64 steps (tests/golden/README.md)."""
import numpy as np
import pytest

from lio_amd import pipeline
from window_util import assert_cost_trace_close, assert_priors_close, assert_windows_close, force_all, make_pair, window_gap

pytestmark = pytest.mark.gpu


def _same_decisions(ra, rb):
    assert ra.iterations == rb.iterations and ra.termination == rb.termination and ra.successful_steps == rb.successful_steps
    assert ra.convergence_flag == rb.convergence_flag and ra.turn_off == rb.turn_off and ra.marginalized == rb.marginalized


def test_vlp16_window15_matches_oracle(hip, oracle):
    """BASELINE.json configs[2]: full sliding-window estimator at VLP-16 size, window 15 / opt window 5."""
    W, Wo = 15, 5
    ds, clouds, (ea, eb) = make_pair((hip, oracle), "indoor", W, Wo, W + 5, 0.2)
    ra, rb = ea.solve(), eb.solve()
    assert rb.n_lidar_residuals > 20000
    _same_decisions(ra, rb)
    gap, flips = assert_cost_trace_close(ra, rb)
    print(f"vlp16 15/5 first solve: trace rel gap {gap:.2e} ({flips} newest-frame factor flips), window gap {window_gap(ea.get_window(), eb.get_window())}")
    assert_windows_close(ea.get_window(), eb.get_window())
    assert_priors_close(ea, eb, ra, rb)
    for est in (ea, eb):
        est.slide()
    force_all(ea, eb, ds)
    for k in range(W + 1, W + 5):
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1])
        rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        _same_decisions(ra, rb)
        assert_cost_trace_close(ra, rb)
        assert_windows_close(ea.get_window(), eb.get_window())
        assert_priors_close(ea, eb, ra, rb)      # the product's OWN marginalization output, every step
        if k < W + 4:
            force_all(ea, eb, ds)
    pa, pb = ea.prior(), eb.prior()
    assert pa["n"] == pb["n"] == 6 * Wo + 15
    rel = np.max(np.abs(pa["JtJ"] - pb["JtJ"])) / np.abs(pb["JtJ"]).max()
    print(f"vlp16 15/5 prior: |dJtJ|/max {rel:.2e}, |dx0| {np.max(np.abs(pa['x0'] - pb['x0'])):.2e}")
    assert rel < 1e-6
    np.testing.assert_allclose(pa["x0"], pb["x0"], atol=1e-4)


@pytest.mark.parametrize("prior_factor", [1, 0])
def test_indoor_12_7_keep_features_chain(hip, oracle, prior_factor):
    """config/indoor_test_config.yaml: window 12 / opt window 7, keep_features 1 (the newest frame's factor list
    accumulates over its <= 10 Gauss-Newton rounds, Estimator.cc:978-980), IMU deskew on.  prior_factor 0 is the shipped
    value; 1 adds the extrinsic prior the other parity tests use.  States are teacher-forced after every step so that a
    one-off borderline feature of the newest frame cannot compound."""
    W, Wo = 12, 7
    ds, clouds, (ea, eb) = make_pair((hip, oracle), "indoor", W, Wo, W + 7, 0.2, keep=1, deskew=True, prior_factor=prior_factor)
    ra, rb = ea.solve(), eb.solve()
    _same_decisions(ra, rb)
    assert rb.laser_odom_iterations >= 2 and ra.laser_odom_iterations == rb.laser_odom_iterations
    # keep_features: the newest frame contributes one factor list per round
    assert rb.n_lidar_residuals > 1.2 * sum(eb.features(f)[0].shape[0] for f in range(W - Wo + 1, W)) / (Wo - 1) * Wo
    assert_cost_trace_close(ra, rb, rtol_floor=2e-4)
    assert_windows_close(ea.get_window(), eb.get_window())
    worst_prior = assert_priors_close(ea, eb, ra, rb, rel_floor=2e-4, tol_ex=1e-4 if prior_factor else 1e-3)[0]   # keep_features: the cost trace's floor
    for est in (ea, eb):
        est.slide()
    force_all(ea, eb, ds)
    worst = 0.0
    for k in range(W + 1, W + 7):
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1])
        rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        _same_decisions(ra, rb)
        assert ra.laser_odom_iterations == rb.laser_odom_iterations
        assert_cost_trace_close(ra, rb, rtol_floor=2e-4)
        wa, wb = ea.get_window(), eb.get_window()
        worst = max(worst, window_gap(wa, wb)[0])
        assert_windows_close(wa, wb)
        # prior_factor 0: the extrinsic is free and weakly observable (window_util.assert_priors_close)
        worst_prior = max(worst_prior, assert_priors_close(ea, eb, ra, rb, rel_floor=2e-4, tol_ex=1e-4 if prior_factor else 1e-3)[0])
        force_all(ea, eb, ds)
    print(f"indoor 12/7 keep_features prior_factor={prior_factor}: worst |dP| over 6 teacher-forced steps {worst:.2e} m, worst |dJtJ|/max of the "
          f"priors {worst_prior:.2e}")
    pa, pb = ea.prior(), eb.prior()
    assert pa["n"] == pb["n"] == 6 * Wo + 15


def test_hdl64_opt_window_15_stress(hip, oracle):
    """The Wo = 15 stress of configs[3]: every frame of the window is optimised (pivot = 0), tangent dimension
    D = 15 (Wo + 1) + 6 = 246, marginalization keeps n = 6 Wo + 15 = 105 columns."""
    W, Wo = 15, 15
    ds, clouds, (ea, eb) = make_pair((hip, oracle), "outdoor", W, Wo, W + 3, 0.3, pp_lib=hip)
    ra, rb = ea.solve(), eb.solve()
    assert rb.n_lidar_residuals > 100000
    _same_decisions(ra, rb)
    assert_cost_trace_close(ra, rb)
    assert_windows_close(ea.get_window(), eb.get_window())
    for est in (ea, eb):
        est.slide()
    force_all(ea, eb, ds)
    for k in range(W + 1, W + 3):
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1])
        rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        _same_decisions(ra, rb)
        print(f"Wo = 15 stress, teacher-forced step {k - W}: window gap {window_gap(ea.get_window(), eb.get_window())}")
        assert_windows_close(ea.get_window(), eb.get_window())     # 1e-4 m / 1e-4 rad
        force_all(ea, eb, ds)
    pa, pb = ea.prior(), eb.prior()
    assert pa["n"] == pb["n"] == 105


@pytest.mark.parametrize("kind,frame_dt", [("indoor", 0.2), ("outdoor", 0.3)])
def test_teacher_forced_chain_of_20_solves(hip, oracle, kind, frame_dt):
    """20 consecutive ProcessLaserOdom calls (push + solve + marginalization + slide) at window 15 / opt window 5 on VLP-16
    and HDL-64E sweeps.  After every step the oracle's window states are copied into the product (lio_est_set_window), so
    each step is compared on the same inputs: 1e-4 m / 1e-4 rad on every step, equal solver decisions, and the clouds the
    two windows carry stay equal to fp32 rounding."""
    W, Wo, n_chain = 15, 5, 20
    ds, clouds, (ea, eb) = make_pair((hip, oracle), kind, W, Wo, W + 1 + n_chain, frame_dt, pp_lib=hip)
    ra, rb = ea.solve(), eb.solve()
    _same_decisions(ra, rb)
    assert_windows_close(ea.get_window(), eb.get_window())
    worst_prior, worst_x0 = assert_priors_close(ea, eb, ra, rb)[:2]
    for est in (ea, eb):
        est.slide()
    force_all(ea, eb, ds)
    worst_p = worst_r = 0.0
    odom_iter_mismatch = 0
    for k in range(W + 1, W + 1 + n_chain):
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1])
        rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        _same_decisions(ra, rb)
        odom_iter_mismatch += int(ra.laser_odom_iterations != rb.laser_odom_iterations)
        wa, wb = ea.get_window(), eb.get_window()
        g = window_gap(wa, wb)
        worst_p, worst_r = max(worst_p, g[0]), max(worst_r, g[1])
        assert_windows_close(wa, wb)
        np.testing.assert_allclose(wa["t_lb"], wb["t_lb"], atol=1e-4)   # the optimised extrinsic travels with the window
        # the prior each side marginalised ITSELF in this step, before the forcing below replaces the product's with the oracle's
        pr = assert_priors_close(ea, eb, ra, rb)
        worst_prior, worst_x0 = max(worst_prior, pr[0]), max(worst_x0, pr[1])
        force_all(ea, eb, ds)
    print(f"teacher-forced chain ({kind}): worst |dP| {worst_p:.2e} m, worst rotation gap {worst_r:.2e} rad over {n_chain} steps; "
          f"{odom_iter_mismatch} steps with a different newest-frame round count; priors: worst |dJtJ|/max {worst_prior:.2e}, |dx0| {worst_x0:.2e}")
    assert odom_iter_mismatch <= 2
    # the clouds never left HBM on the product side: after 20 slides they still match the oracle's
    for f in (W - Wo, W - 1, W):
        sa, sb = ea.get_surf_stack(f), eb.get_surf_stack(f)
        assert sa.shape == sb.shape
        np.testing.assert_allclose(sa[:, :3], sb[:, :3], atol=5e-4)


def test_stream_sync_fallback_gives_the_same_bits(hip, oracle):
    """lio_est_config.stream_sync = 1 (the LIO_HOST_SIGNAL=0 path: D2H copies + hipStreamSynchronize instead of completion words in
    host memory) must stay alive: same kernels, same arithmetic => the same window bit for bit over solve + slide + solve.  This
    also pins the resident moments kernel to its launch form: with stream_sync the lidar moments come from k_lidar_moments +
    k_moment_reduce launches instead of the resident kernel's passes."""
    from lio_amd import capi, synth
    W, Wo = 8, 4
    ds = synth.make_dataset("indoor", W + 3, 0.2)
    clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
    wins = []
    for sync in (0, 1):
        cfg = pipeline.config_indoor(hip, W, Wo)
        cfg.keep_features, cfg.cutoff_deskew, cfg.prior_factor, cfg.stream_sync = 0, 1, 1, sync
        pipeline.set_extrinsic(cfg, ds)
        est = capi.Estimator(hip, cfg)
        pipeline.init_window(est, hip, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01)
        reps = [est.solve()]
        est.slide()
        for k in (W + 1, W + 2):
            reps.append(pipeline.feed_frame(est, ds, k, clouds[k][0], clouds[k][1]))
        wins.append((est.get_window(), [r.final_cost for r in reps], [r.n_lidar_residuals for r in reps]))
    (wa, ca, na), (wb, cb, nb) = wins
    assert na == nb and ca == cb
    for key in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
        np.testing.assert_array_equal(wa[key], wb[key])


@pytest.mark.parametrize("kind,W,Wo,frame_dt", [("indoor", 8, 4, 0.2), ("outdoor", 15, 5, 0.3)])
def test_resident_moments_equal_the_launch_pair(hip, kind, W, Wo, frame_dt):
    """the resident moments kernel (one launch per solve, one pass per linearisation, doorbell in host memory) and
    the k_lidar_moments + k_moment_reduce launch pair over the SAME partition of the factor slots (lio_est_config.resident_moments
    = 3: what a solve gets when the resident form is refused — another solve in flight, factor sharding, stream_sync) share the
    per-block arithmetic and the fold order, so the solve is the same bit for bit: cost trace, decisions, window — at VLP-16 size
    and on the HDL-64 headline window.  The pass counter says which of the two actually ran (a silent fallback would make the
    comparison empty).  Round 2's launch pair (resident_moments = 2) partitions the slots differently: the same sums in another
    order, compared at 1e-9."""
    from lio_amd import capi, synth
    ds = synth.make_dataset(kind, W + 2, frame_dt)
    clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
    runs = {}
    for resident in (1, 3, 2):
        cfg = pipeline.config_indoor(hip, W, Wo) if kind == "indoor" else pipeline.config_outdoor64(hip, W, Wo)
        # keep_features would give the newest frame rounds x M factor slots: 10 rounds x 6.5 k slots need more than the 256 co-resident
        # blocks of the resident form and the solve takes the launch pairs (same bits; nothing to compare)
        cfg.prior_factor, cfg.keep_features, cfg.resident_moments = 1, 0, resident
        pipeline.set_extrinsic(cfg, ds)
        est = capi.Estimator(hip, cfg)
        pipeline.init_window(est, hip, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01)
        reps = [est.solve()]
        est.slide()
        reps.append(pipeline.feed_frame(est, ds, W + 1, clouds[W + 1][0], clouds[W + 1][1]))
        passes = est.kernel_timing("moments_resident")["launches"]
        evaluations = sum(r.iterations + 1 for r in reps)
        if resident == 1:
            assert passes >= evaluations, (passes, evaluations)   # every linearisation of both solves was a pass of the resident kernel
        else:
            assert passes == 0
        traces = [list(r.cost_trace[:r.iterations + 1]) for r in reps]
        runs[resident] = (est.get_window(), traces, [(r.iterations, r.termination, r.successful_steps, r.n_lidar_residuals) for r in reps])
    (wa, ta, da), (wb, tb, db), (wc, tc, dc) = runs[1], runs[3], runs[2]
    assert da == db and ta == tb
    for key in ("Ps", "Rs", "Vs", "Bas", "Bgs", "t_lb"):
        np.testing.assert_array_equal(wa[key], wb[key])
    assert da == dc
    for x, y in zip(ta, tc):
        np.testing.assert_allclose(x, y, rtol=1e-9)
    for key in ("Ps", "Rs", "Vs", "Bas", "Bgs", "t_lb"):
        np.testing.assert_allclose(wa[key], wc[key], atol=1e-8)
