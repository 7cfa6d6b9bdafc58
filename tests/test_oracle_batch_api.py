"""CPU checks of the batch entry points through the ORACLE's implementation of the same header (include/lio_c.h): lio_est_batch_*
(the oracle solves the members one after the other — what the product's batch must equal window by window, tests/test_gpu_batch.py)
and lio_pp_process_batch.  What is checked here is the contract both libraries share: adoption and release of handles, argument
errors, reports in member order, a batch step equal to the members' own steps."""
import numpy as np
import pytest

from lio_amd import capi, pipeline, synth


def _estimators(oracle, seeds, W=4, Wo=2):
    ds = synth.make_dataset("indoor", W + 3, 0.2)
    clouds = [pipeline.feature_clouds(oracle, ds.lidar, f.scan) for f in ds.frames]
    out = []
    for s in seeds:
        cfg = pipeline.config_indoor(oracle, W, Wo)
        cfg.keep_features, cfg.prior_factor, cfg.cutoff_deskew = 0, 1, 1
        pipeline.set_extrinsic(cfg, ds)
        e = capi.Estimator(oracle, cfg)
        pipeline.init_window(e, oracle, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=s)
        out.append(e)
    return out


def test_batch_of_the_oracle_equals_its_members(oracle):
    solo = _estimators(oracle, (3, 11))
    members = _estimators(oracle, (3, 11))
    batch = capi.EstimatorBatch(oracle, members)
    assert len(batch) == 2
    reps = batch.solve()
    for e, m, rb in zip(solo, members, reps):
        ra = e.solve()
        assert (ra.iterations, ra.termination, ra.n_lidar_residuals, ra.final_cost) == (rb.iterations, rb.termination, rb.n_lidar_residuals, rb.final_cost)
        wa, wb = e.get_window(), m.get_window()
        for key in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
            np.testing.assert_array_equal(wa[key], wb[key])
    batch.close()


def test_batch_arguments(oracle):
    a, b = _estimators(oracle, (1, 2))
    with pytest.raises(capi.LioError):
        capi.EstimatorBatch(oracle, [a, a])            # a window twice
    with pytest.raises(capi.LioError):
        capi.EstimatorBatch(oracle, [])                # no window
    batch = capi.EstimatorBatch(oracle, [a, b])
    with pytest.raises(capi.LioError):
        capi.EstimatorBatch(oracle, [b])               # already adopted
    batch.close()
    capi.EstimatorBatch(oracle, [a]).close()           # released by the batch that is gone: adoptable again


def test_batch_contract_shared_with_the_product(oracle):
    """What include/lio_c.h promises on the error paths, identically in both libraries (the product's side: tests/test_gpu_batch.py::
    test_batch_arguments): an uninitialised member -> LIO_ERR_STATE before ANY window is solved; options by name; a member destroyed
    before its batch dissolves the batch instead of leaving it with a dangling pointer."""
    a, b = _estimators(oracle, (1, 2))
    ds = synth.make_dataset("indoor", 7, 0.2)
    cfg = pipeline.config_indoor(oracle, 4, 2)
    pipeline.set_extrinsic(cfg, ds)
    fresh = capi.Estimator(oracle, cfg)                # never initialised
    before = a.get_window()
    batch = capi.EstimatorBatch(oracle, [a, fresh, b])
    with pytest.raises(capi.LioError):
        batch.solve()
    for key in ("Ps", "Rs", "Vs"):
        np.testing.assert_array_equal(a.get_window()[key], before[key])   # member 0 was NOT solved
    batch.set_option("lanes_per_query", 1)
    batch.set_option("loop_groups", 2)
    with pytest.raises(capi.LioError):
        batch.set_option("no_such_option", 1)
    # destroying a member first: the batch is dissolved, its other members are free again, the batch handle only accepts destroy
    oracle.dll.lio_est_destroy(fresh.h)
    fresh.h = None
    with pytest.raises(capi.LioError):
        batch.solve()
    assert len(batch) == 0
    other = capi.EstimatorBatch(oracle, [a, b])        # released
    assert len(other.solve()) == 2
    other.close()
    batch.close()


def test_point_processor_batch_equals_one_by_one(oracle):
    ds = synth.make_dataset("indoor", 3, 0.1)
    lid = ds.lidar
    batch = [capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings) for _ in ds.frames]
    single = [capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings) for _ in ds.frames]
    capi.PointProcessor.process_batch(batch, [f.scan for f in ds.frames])
    for p, f in zip(single, ds.frames):
        p.process(f.scan)
    for a, b in zip(batch, single):
        for which in range(5):
            np.testing.assert_array_equal(a.cloud(which), b.cloud(which))
    with pytest.raises(capi.LioError):
        capi.PointProcessor.process_batch([batch[0], batch[0]], [ds.frames[0].scan] * 2)
    capi.PointProcessor.process_batch([], [])          # nothing to do is not an error
    # lio_pp_process_batch_device through the oracle: it has no device, so the "device" pointers are host memory — same results
    again = [capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings) for _ in ds.frames]
    arrs = [np.ascontiguousarray(f.scan, np.float32) for f in ds.frames]
    capi.PointProcessor.process_batch_device(again, [a.ctypes.data for a in arrs], [a.shape[0] for a in arrs])
    for a, b in zip(again, single):
        for which in range(5):
            np.testing.assert_array_equal(a.cloud(which), b.cloud(which))
        np.testing.assert_array_equal(a.indices(2)[1], b.indices(2)[1])
