"""lio_est_batch at the sizes bench.py runs (the batched figures had no parity evidence — the largest batch any test
built was three headline windows, and bench.py's own `all_windows_same_decisions` was false at 64 windows).

Root cause found in round 6 (profiles/r6_a_determinism_diagnosis.txt): two races on the step kernel's LDS control block (solve_step.h).
They only bit when blocks of the OTHER loop group shared a SIMD with a step workgroup — i.e. from 32 windows on, at random.

* every kernel variant and host path the launch sizes select (one / two / four / eight lanes per query, the forced-occupancy forms,
  one to three loop groups, the aux row's block sizes, the threaded write-back) is forced on a small batch through
  lio_est_batch_set_option: same bits in every stage (lio_est_batch_stage_digest), same reports, windows and priors;
* B identical copies of BASELINE.json's headline window (HDL-64E, window 15 / opt 5) at B = 40 (k_bw_features1_w8,
  k_bw_odom_round<1>, two loop groups) and B = 136 (+ one-wave aux blocks, four write-back threads): every copy equals copy 0 in
  every stage on repeated steps, copy 0 equals the window solved alone (a batch of one) bit for bit and the oracle within
  1e-4 m / 1e-4 rad at equal iteration counts; then one more frame WITHOUT a restore, so that the device-resident prior feeds the
  next solve at scale."""
import numpy as np
import pytest

from lio_amd import capi, pipeline, synth
from window_util import assert_windows_close, window_gap

pytestmark = pytest.mark.gpu

STAGES = range(len(capi.EstimatorBatch.STAGES))


def _cfg(lib, kind, W, Wo, keep=0, opt_extrinsic=0):
    cfg = pipeline.config_indoor(lib, W, Wo) if kind == "indoor" else pipeline.config_outdoor64(lib, W, Wo)
    cfg.keep_features, cfg.prior_factor, cfg.cutoff_deskew, cfg.opt_extrinsic = keep, 1, 1, opt_extrinsic
    return cfg


def _push(est, ds, k, surf, corner):
    f = ds.frames[k]
    for j in range(f.imu_dt.shape[0]):
        est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
    est.push_frame(capi.TransformF.make([0, 0, 0, 1], [0, 0, 0]), surf, corner, f.t)


def _rep_key(r):
    return (r.iterations, r.successful_steps, r.termination, r.n_lidar_residuals, r.n_local_map, r.laser_odom_iterations, r.turn_off, r.convergence_flag,
            r.marginalized, r.initial_cost, r.final_cost, tuple(r.cost_trace[:12]))


def _digests(batch):
    return np.stack([batch.stage_digest(s) for s in STAGES])     # (stages, windows)


def _state(batch, reps):
    out = []
    for e, r in zip(batch.members, reps):
        w = e.get_window()
        p = e.prior()
        out.append((_rep_key(r), {k: w[k].copy() for k in ("Ps", "Rs", "Vs", "Bas", "Bgs", "q_lb", "t_lb")}, None if p is None else p["JtJ"].copy()))
    return out


def _assert_same_state(a, b, what):
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0], (what, k, x[0], y[0])
        for key in x[1]:
            np.testing.assert_array_equal(x[1][key], y[1][key], err_msg=f"{what}: window {k} {key}")
        assert (x[2] is None) == (y[2] is None)
        if x[2] is not None:
            np.testing.assert_array_equal(x[2], y[2], err_msg=f"{what}: window {k} prior")


OPTION_SETS = [   # (lanes_per_query, occupancy, loop_groups, aux_threads, finish_threads, parts)
    (8, -1, 1, 256, 1, 1),       # what a small batch takes by itself
    (4, -1, 1, 256, 1, 2),       # ... as two parts on two host threads (2 + 1 windows)
    (2, -1, 2, 128, 2, 1),
    (1, 0, 2, 64, 1, 2),         # one lane per query as compiled
    (1, 6, 3, 64, 3, 1),
    (1, 8, 0, 64, 8, 2),         # the forms a batch of 512 headline windows runs
    (0, -1, 0, 0, 0, 0),         # everything by size again
]


def test_execution_choices_do_not_change_a_bit(hip):
    """Three different windows (sizes, keep_features) in one batch, restored and solved under every option set, then pushed one frame
    further without a restore (the prior comes from the device): digests of all stages, reports, windows and priors are the first set's."""
    specs = [("indoor", 4, 2, 0, 11, 3), ("indoor", 6, 3, 1, 13, 5), ("indoor", 5, 2, 0, 12, 7)]
    runs = []
    for kind, W, Wo, keep, nfr, seed in specs:
        ds = synth.make_dataset(kind, nfr, 0.2)
        clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
        cfg = _cfg(hip, kind, W, Wo, keep)
        pipeline.set_extrinsic(cfg, ds)
        est = capi.Estimator(hip, cfg)
        pipeline.init_window(est, hip, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=seed)
        runs.append((ds, clouds, W, est))
    batch = capi.EstimatorBatch(hip, [r[3] for r in runs])
    # two solves + slides bring every window onto the device loop with a prior; the state to compare from is snapshotted there
    for step in range(2):
        batch.solve()
        for ds, clouds, W, est in runs:
            est.slide()
            _push(est, ds, W + 1 + step, clouds[W + 1 + step][0], clouds[W + 1 + step][1])
    for _, _, _, est in runs:
        est.snapshot()
    first = None
    for lpq, occ, groups, aux, fin, parts in OPTION_SETS:
        for name, v in (("parts", parts), ("lanes_per_query", lpq), ("occupancy", occ), ("loop_groups", groups), ("aux_threads", aux), ("finish_threads", fin)):
            batch.set_option(name, v)
        reps = batch.solve_restored(1)
        assert int(batch.clock()["n_device"]) == len(runs)
        dg = _digests(batch)
        st1 = _state(batch, reps)
        for ds, clouds, W, est in runs:       # one more frame, no restore: the next solve's prior is the one the device has just made
            est.slide()
            _push(est, ds, W + 3, clouds[W + 3][0], clouds[W + 3][1])
        reps2 = batch.solve()
        dg2 = _digests(batch)
        st2 = _state(batch, reps2)
        if first is None:
            first = (dg, st1, dg2, st2)
            assert all(k[0][8] == 1 for k in st1), "the compared step must marginalise (the prior of the second step comes from the device)"
            continue
        what = f"options {(lpq, occ, groups, aux, fin, parts)}"
        np.testing.assert_array_equal(dg, first[0], err_msg=what)
        np.testing.assert_array_equal(dg2, first[2], err_msg=what + " (second step)")
        _assert_same_state(st1, first[1], what)
        _assert_same_state(st2, first[3], what + " (second step)")
    with pytest.raises(capi.LioError):
        batch.set_option("lanes_per_query", 3)
    batch.close()


@pytest.fixture(scope="module")
def headline(hip):
    W, Wo = 15, 5
    ds = synth.make_dataset("outdoor", W + 2, 0.3)
    clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
    return W, Wo, ds, clouds


def _headline_estimator(lib, headline, seed=3):
    W, Wo, ds, clouds = headline
    cfg = _cfg(lib, "outdoor", W, Wo)
    pipeline.set_extrinsic(cfg, ds)
    est = capi.Estimator(lib, cfg)
    pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=seed)
    return est


@pytest.mark.parametrize("B", [40, 136])
def test_identical_windows_agree_at_scale(hip, oracle, headline, B):
    W, Wo, ds, clouds = headline
    est0 = _headline_estimator(hip, headline)
    est0.snapshot()
    cfg = _cfg(hip, "outdoor", W, Wo)
    pipeline.set_extrinsic(cfg, ds)
    clones = []
    for _ in range(B):
        e = capi.Estimator(hip, cfg)
        e.copy_snapshot_of(est0)
        e.restore()
        clones.append(e)
    # ---- the window alone: a batch of one
    alone = capi.Estimator(hip, cfg)
    alone.copy_snapshot_of(est0)
    alone.restore()
    b1 = capi.EstimatorBatch(hip, [alone])
    rep1 = b1.solve_restored(1)[0]
    dg1 = _digests(b1)[:, 0]
    # ---- B copies, repeated steps (the round-5 failure was intermittent: about every second step at 64 windows)
    batch = capi.EstimatorBatch(hip, clones)
    for trial in range(4):
        reps = batch.solve_restored(1 if trial else 2)
        clk = batch.clock()
        assert int(clk["n_device"]) == B
        dg = _digests(batch)
        for s in STAGES:
            bad = np.nonzero(dg[s] != dg[s][0])[0]
            assert bad.size == 0, f"trial {trial}: stage {capi.EstimatorBatch.STAGES[s]} of windows {bad[:10].tolist()} differs from window 0"
        np.testing.assert_array_equal(dg[:, 0], dg1, err_msg=f"trial {trial}: window 0 of {B} against the window alone")
        for r in reps:
            assert _rep_key(r) == _rep_key(rep1)
    w1 = alone.get_window()
    for e in (clones[0], clones[B // 2], clones[-1]):
        wb = e.get_window()
        for key in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
            np.testing.assert_array_equal(wb[key], w1[key])
    # ---- against the oracle on the same window
    eo = _headline_estimator(oracle, headline)
    ro = eo.solve()
    assert reps[0].iterations == ro.iterations and reps[0].termination == ro.termination
    assert abs(reps[0].n_lidar_residuals - ro.n_lidar_residuals) <= 0.002 * ro.n_lidar_residuals
    assert_windows_close(clones[0].get_window(), eo.get_window())
    gap = window_gap(clones[0].get_window(), eo.get_window())[0]
    # ---- one more frame without a restore: the prior of this solve is the one the batch's marginalization left on the device
    k = W + 1
    for e in clones + [alone]:
        e.slide()
        _push(e, ds, k, clouds[k][0], clouds[k][1])
    reps2 = batch.solve()
    rep2 = b1.solve()[0]
    assert int(batch.clock()["n_device"]) == B
    dg2, dg2_1 = _digests(batch), _digests(b1)[:, 0]
    for s in STAGES:
        bad = np.nonzero(dg2[s] != dg2[s][0])[0]
        assert bad.size == 0, f"second step: stage {capi.EstimatorBatch.STAGES[s]} of windows {bad[:10].tolist()} differs from window 0"
    np.testing.assert_array_equal(dg2[:, 0], dg2_1)
    assert all(_rep_key(r) == _rep_key(rep2) for r in reps2) and rep2.marginalized == 1
    pa = alone.prior()
    for e in (clones[0], clones[-1]):
        np.testing.assert_array_equal(e.prior()["JtJ"], pa["JtJ"])
    print(f"{B} copies of the headline window: all stages equal over 4 steps + an un-restored step, = the window alone; vs oracle |dP| {gap:.2e} m, "
          f"{reps[0].iterations} iterations, {reps[0].n_lidar_residuals} residuals; features kernel by size at {B} windows")
    batch.close()
    b1.close()
