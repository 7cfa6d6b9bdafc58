"""lio_est_batch: B windows per launch chain — csrc/est_batch.hip, batch_kernels.hip, the batched launch A /
launch B of solve_kernels.hip and the batched marginalization of marg_kernels.hip.

* every window of a batch is BIT-IDENTICAL to the same window solved alone (a batch of one = lio_est_config.device_solve), over
  chains that pass through the single-window fallback (convergence_flag_ not set yet) and through the device loop + device
  marginalization, with different window sizes, keep_features on and off, in one batch;
* a batch against the oracle, teacher-forced step by step: equal decisions and iteration counts, 1e-4 m / 1e-4 rad;
* BASELINE.json's HDL-64E / window 15 / opt 5 configuration in a batch, against the oracle and against the single-window handle."""
import numpy as np
import pytest

from lio_amd import capi, pipeline, synth
from window_util import assert_priors_close, assert_windows_close, force_all, window_gap

pytestmark = pytest.mark.gpu


def _cfg(lib, kind, W, Wo, keep, opt_extrinsic=1, device_solve=0):
    cfg = pipeline.config_indoor(lib, W, Wo) if kind == "indoor" else pipeline.config_outdoor64(lib, W, Wo)
    cfg.keep_features, cfg.prior_factor, cfg.cutoff_deskew, cfg.opt_extrinsic = keep, 1, 1, opt_extrinsic
    cfg.device_solve = device_solve
    return cfg


def _push(est, ds, k, surf, corner):
    """everything of ProcessLaserOdom in front of SolveOptimization (Estimator.cc:441-488, 620-693)"""
    f = ds.frames[k]
    for j in range(f.imu_dt.shape[0]):
        est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
    est.push_frame(capi.TransformF.make([0, 0, 0, 1], [0, 0, 0]), surf, corner, f.t)


def _rep_key(r):
    return (r.iterations, r.successful_steps, r.termination, r.n_lidar_residuals, r.n_local_map, r.laser_odom_iterations, r.turn_off, r.convergence_flag,
            r.marginalized, r.initial_cost, r.final_cost, tuple(r.cost_trace[:12]))


SPECS = [  # kind, W, Wo, keep_features, opt_extrinsic, frames, seed
    ("indoor", 4, 2, 0, 1, 11, 3),
    ("indoor", 6, 3, 1, 1, 13, 5),
    ("indoor", 5, 2, 0, 0, 12, 7),      # constant extrinsic: the device loop takes the very first solve (no prior yet)
]


def test_every_window_of_a_batch_equals_the_window_alone(hip):
    """Three different windows (sizes, keep_features, extrinsic handling) solved (a) each alone through a batch of one and (b) together
    in one batch, over solve + slide + five more frames: same reports, same windows, same priors, bit for bit."""
    runs = []
    for kind, W, Wo, keep, optex, nfr, seed in SPECS:
        ds = synth.make_dataset(kind, nfr, 0.2)
        clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
        pair = []
        for device_solve in (1, 0):
            cfg = _cfg(hip, kind, W, Wo, keep, optex, device_solve)
            pipeline.set_extrinsic(cfg, ds)
            est = capi.Estimator(hip, cfg)
            pipeline.init_window(est, hip, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=seed)
            pair.append(est)
        runs.append((ds, clouds, W, pair[0], pair[1]))
    batch = capi.EstimatorBatch(hip, [r[4] for r in runs])
    assert len(batch) == len(SPECS)
    on_device = 0
    for step in range(6):
        if step > 0:
            for ds, clouds, W, solo, member in runs:
                k = W + step
                _push(solo, ds, k, clouds[k][0], clouds[k][1])
                _push(member, ds, k, clouds[k][0], clouds[k][1])
        solo_reps = [r[3].solve() for r in runs]
        batch_reps = batch.solve()
        on_device += int(batch.clock()["n_device"])
        for (ds, clouds, W, solo, member), ra, rb in zip(runs, solo_reps, batch_reps):
            assert _rep_key(ra) == _rep_key(rb), (step, _rep_key(ra), _rep_key(rb))
            wa, wb = solo.get_window(), member.get_window()
            for key in ("Ps", "Rs", "Vs", "Bas", "Bgs", "q_lb", "t_lb"):
                np.testing.assert_array_equal(wa[key], wb[key], err_msg=f"step {step} {key}")
            solo.slide(); member.slide()
    for ds, clouds, W, solo, member in runs:
        pa, pb = solo.prior(), member.prior()
        assert (pa is None) == (pb is None)
        if pa is not None:
            for key in ("JtJ", "Jtr", "x0"):
                np.testing.assert_array_equal(pa[key], pb[key])
    print(f"batch == solo over 6 steps x {len(SPECS)} windows; {on_device} of {6 * len(SPECS)} window-solves ran the device loop (the rest: single-window fallback)")
    assert on_device >= 2 * len(SPECS)
    batch.close()


# (10 / 6: the largest problem the device loop takes — 111 unknowns in a 112 x 112 H in LDS, a 51-column prior.  12 / 7: the reference's
# indoor_test_config.yaml — 126 unknowns do not fit the LDS-resident factorisation: the batch solves such windows by the single-window
# path inside the same call, same parity bar)
@pytest.mark.parametrize("kind,W,Wo,frame_dt,min_device", [("indoor", 4, 2, 0.2, 4), ("outdoor", 6, 3, 0.3, 4), ("indoor", 10, 6, 0.2, 4), ("indoor", 12, 7, 0.2, 0)])
def test_batch_matches_the_oracle(hip, oracle, kind, W, Wo, frame_dt, min_device):
    """Two windows of one batch (the same scene from two perturbed starts) against two oracle estimators, teacher-forced before
    every step (states, extrinsic, prior): equal convergence flags and iteration counts, windows within 1e-4 m / 1e-4 rad on every
    step, the priors each side produced itself within 1e-6 relative."""
    nfr = W + 7
    ds = synth.make_dataset(kind, nfr, frame_dt)
    clouds = [pipeline.feature_clouds(oracle, ds.lidar, f.scan) for f in ds.frames]
    prod, orc = [], []
    for seed in (3, 11):
        for lib, dst in ((hip, prod), (oracle, orc)):
            cfg = _cfg(lib, kind, W, Wo, 0)
            pipeline.set_extrinsic(cfg, ds)
            est = capi.Estimator(lib, cfg)
            pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=seed)
            dst.append(est)
    batch = capi.EstimatorBatch(hip, prod)
    worst, on_device = 0.0, 0
    for step in range(6):
        if step > 0:
            k = W + step
            for ea, eb in zip(prod, orc):
                force_all(ea, eb, ds)
                _push(ea, ds, k, clouds[k][0], clouds[k][1])
                _push(eb, ds, k, clouds[k][0], clouds[k][1])
        ra = batch.solve()
        rb = [e.solve() for e in orc]
        on_device += int(batch.clock()["n_device"])
        for ea, eb, a, b in zip(prod, orc, ra, rb):
            assert (a.convergence_flag, a.turn_off, a.marginalized) == (b.convergence_flag, b.turn_off, b.marginalized), step
            assert a.iterations == b.iterations and a.termination == b.termination, (step, a.iterations, b.iterations)
            assert abs(a.n_lidar_residuals - b.n_lidar_residuals) <= max(5, 0.002 * b.n_lidar_residuals)
            assert_windows_close(ea.get_window(), eb.get_window())
            worst = max(worst, window_gap(ea.get_window(), eb.get_window())[0])
            if step > 0:
                assert_priors_close(ea, eb, a, b, rel_floor=1e-5)
            ea.slide(); eb.slide()
    print(f"batch vs oracle ({kind} {W}/{Wo}): worst |dP| over the teacher-forced steps {worst:.2e} m; {on_device} window-solves on the device loop")
    assert on_device >= min_device
    if min_device == 0:
        assert on_device == 0          # (every solve of these windows went through the single-window path)
    batch.close()


def test_headline_configuration_in_a_batch(hip, oracle):
    """BASELINE.json configs[3]: HDL-64E sweeps (~133 k points), window 15 / opt window 5, three windows in one batch (restored from
    their snapshots, the bench's step): against the oracle on the same window, against the single-window handle of the default
    path, and each window against itself solved alone."""
    W, Wo = 15, 5
    ds = synth.make_dataset("outdoor", W + 2, 0.3)
    clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]

    def make(lib, seed, device_solve=0, optex=0):
        cfg = _cfg(lib, "outdoor", W, Wo, 0, optex, device_solve)
        pipeline.set_extrinsic(cfg, ds)
        est = capi.Estimator(lib, cfg)
        pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=seed)
        return est

    seeds = (3, 4, 5)
    members = [make(hip, s) for s in seeds]
    for e in members:
        e.snapshot()
    batch = capi.EstimatorBatch(hip, members)
    reps = batch.solve_restored(2)
    clk = batch.clock()
    assert int(clk["n_device"]) == len(seeds), clk
    print("batch clock (ms):", {k: round(v, 3) for k, v in clk.items()})
    # the oracle and the default single-window path on the first window
    eo, eh = make(oracle, seeds[0]), make(hip, seeds[0])
    ro, rh = eo.solve(), eh.solve()
    r0 = reps[0]
    assert r0.n_lidar_residuals > 30000
    assert r0.iterations == ro.iterations == rh.iterations and r0.termination == ro.termination
    assert abs(r0.n_lidar_residuals - ro.n_lidar_residuals) <= 0.002 * ro.n_lidar_residuals
    assert_windows_close(members[0].get_window(), eo.get_window())
    assert_windows_close(members[0].get_window(), eh.get_window())
    print(f"headline window in a batch: vs oracle |dP| {window_gap(members[0].get_window(), eo.get_window())[0]:.2e} m, vs the single-window path "
          f"{window_gap(members[0].get_window(), eh.get_window())[0]:.2e} m; {r0.n_lidar_residuals} residuals, {r0.iterations} iterations")
    # alone == in the batch, over this solve and the next frame's (which marginalises: the prior comes from the device)
    k = W + 1
    solos = []
    for s, m, rb in zip(seeds, members, reps):
        solo = make(hip, s, device_solve=1)
        ra = solo.solve()
        assert _rep_key(ra) == _rep_key(rb)
        solos.append(solo)
    for e in solos + members:
        e.slide()
        _push(e, ds, k, clouds[k][0], clouds[k][1])
    ra2 = [e.solve() for e in solos]
    rb2 = batch.solve()
    assert int(batch.clock()["n_device"]) == len(seeds)
    for solo, m, ra, rb in zip(solos, members, ra2, rb2):
        assert _rep_key(ra) == _rep_key(rb)
        wa, wb = solo.get_window(), m.get_window()
        for key in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
            np.testing.assert_array_equal(wa[key], wb[key])
        pa, pb = solo.prior(), m.prior()
        assert (pa is None) == (pb is None)
        if pa is not None:
            np.testing.assert_array_equal(pa["JtJ"], pb["JtJ"])
    print("second step:", [(r.iterations, r.marginalized, r.n_lidar_residuals) for r in rb2])
    batch.close()


def test_batch_arguments(hip):
    ds = synth.make_dataset("indoor", 6, 0.2)
    cfg = _cfg(hip, "indoor", 4, 2, 0)
    pipeline.set_extrinsic(cfg, ds)
    a, b = capi.Estimator(hip, cfg), capi.Estimator(hip, cfg)
    with pytest.raises(capi.LioError):
        capi.EstimatorBatch(hip, [a, a])          # a window twice
    batch = capi.EstimatorBatch(hip, [a, b])
    with pytest.raises(capi.LioError):
        capi.EstimatorBatch(hip, [b])             # already adopted
    with pytest.raises(capi.LioError):
        batch.solve()                             # not initialised
    batch.set_option("loop_groups", 2)
    with pytest.raises(capi.LioError):
        batch.set_option("no_such_option", 1)
    with pytest.raises(capi.LioError):
        batch.set_option("aux_threads", 100)
    batch.close()
    capi.EstimatorBatch(hip, [a]).close()         # released by the batch that is gone: adoptable again
    # a member destroyed before its batch: the batch is dissolved (no dangling pointer), the other member is free again
    batch = capi.EstimatorBatch(hip, [a, b])
    hip.dll.lio_est_destroy(b.h)
    b.h = None
    assert len(batch) == 0
    with pytest.raises(capi.LioError):
        batch.solve()
    capi.EstimatorBatch(hip, [a]).close()
    batch.close()


def test_aux_row_does_not_depend_on_its_block_size(hip):
    """launch A's aux row (IMU factors, priors) runs 256 threads per block in small launches and one wave per block from 128 windows
    per launch on: the same window solved with either (lio_est_batch_set_option "aux_threads") gives the same bits — what makes a
    window in a batch of 512 equal to the window alone.  (All execution choices together: tests/test_gpu_batch_scale.py.)"""
    kind, W, Wo = "indoor", 5, 2
    ds = synth.make_dataset(kind, W + 4, 0.2)
    clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
    cfg = _cfg(hip, kind, W, Wo, 0, 0)
    pipeline.set_extrinsic(cfg, ds)
    est = capi.Estimator(hip, cfg)
    pipeline.init_window(est, hip, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=5)
    batch = capi.EstimatorBatch(hip, [est])
    batch.solve(); est.slide()
    _push(est, ds, W + 1, clouds[W + 1][0], clouds[W + 1][1])
    batch.solve(); est.slide()                                   # (the device loop has a prior from here on)
    _push(est, ds, W + 2, clouds[W + 2][0], clouds[W + 2][1])
    est.snapshot()
    got = []
    for threads in (256, 64, 128):
        batch.set_option("aux_threads", threads)
        rep = batch.solve_restored(1)[0]
        got.append((_rep_key(rep), est.get_window(), est.prior()))
    assert got[0][0][0] >= 1
    for other in got[1:]:
        assert other[0] == got[0][0]
        for key in ("Ps", "Rs", "Vs", "Bas", "Bgs", "q_lb", "t_lb"):
            np.testing.assert_array_equal(other[1][key], got[0][1][key])
        assert (other[2] is None) == (got[0][2] is None)
        if other[2] is not None:
            np.testing.assert_array_equal(other[2]["JtJ"], got[0][2]["JtJ"])
    batch.close()


def test_window_beyond_the_filter_s_key_range_takes_the_single_window_path(hip, oracle):
    """The batched VoxelGrid keys a point by 31 bits of absolute cell coordinates (z: +-255 cells); a window with a return far above
    that (a stray point 300 m up) is handed to the single-window path inside the same call: the batch reports it as solved on the
    host side, its neighbour in the batch stays on the device loop, and both equal what the oracle computes for them."""
    kind, W, Wo = "indoor", 4, 2
    ds = synth.make_dataset(kind, W + 3, 0.2)
    clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
    prod, orc = [], []
    for lib, dst in ((hip, prod), (oracle, orc)):
        for seed in (3, 5):
            cfg = _cfg(lib, kind, W, Wo, 0, 0)
            pipeline.set_extrinsic(cfg, ds)
            e = capi.Estimator(lib, cfg)
            pipeline.init_window(e, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=seed)
            dst.append(e)
    batch = capi.EstimatorBatch(hip, prod)
    for step in range(3):
        if step > 0:
            k = W + step
            for ea, eb in zip(prod, orc):
                force_all(ea, eb, ds)
                _push(ea, ds, k, clouds[k][0], clouds[k][1])
                _push(eb, ds, k, clouds[k][0], clouds[k][1])
        if step == 2:                                        # the stray return goes into the pivot frame's stack of window 0, on both sides
            pivot = W - Wo
            stack = prod[0].get_surf_stack(pivot)
            stray = np.vstack([stack, np.array([[0.5, 0.5, 300.0, stack[0, 3]]], np.float32)])
            prod[0].set_surf_stack(pivot, stray)
            orc[0].set_surf_stack(pivot, stray)
        ra = batch.solve()
        rb = [e.solve() for e in orc]
        clk = batch.clock()
        if step == 2:
            assert int(clk["n_device"]) == 1, clk            # window 1 on the device loop, window 0 handed over
        for ea, eb, a, b in zip(prod, orc, ra, rb):
            assert a.iterations == b.iterations and a.termination == b.termination, (step, a.iterations, b.iterations)
            assert_windows_close(ea.get_window(), eb.get_window())
            ea.slide(); eb.slide()
    batch.close()
