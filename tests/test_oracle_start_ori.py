"""CPU checks of config_.infer_start_ori_ (PointProcessor.cc:348-387) as the oracle restates it, against a float64 model
written from the description of the filter (tests/start_ori_util.py)."""
import numpy as np

from lio_amd import capi, synth
from start_ori_util import FilterModel, azimuth, make_sweeps


def _cfg(lib, infer, rad_diff=0.2):
    cfg = capi.PPConfig()
    lib.dll.lio_pp_default_config(cfg)
    assert cfg.infer_start_ori == 0 and cfg.rad_diff == 0.2          # PointProcessor.h:117-119
    cfg.infer_start_ori, cfg.rad_diff = int(infer), rad_diff
    return cfg


def _ring0_front(pp):
    off = pp.ring_offsets()
    return off[1] > off[0]


def test_filter_replaces_a_jump_and_recovers(oracle):
    ds = synth.make_dataset("indoor", 1, 0.1)
    lid = ds.lidar
    stray = (14, 15, 22)
    sweeps = make_sweeps(ds.frames[0].scan, 30, 0.03, stray_at=stray)
    plain = capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings, _cfg(oracle, False))
    infer = capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings, _cfg(oracle, True))
    assert np.isnan(infer.start_ori())
    model = FilterModel(0.2)
    replaced = 0
    for k, scan in enumerate(sweeps):
        plain.process(scan)
        infer.process(scan)
        measured = azimuth(float(scan[0, 0]), float(scan[0, 1]))
        assert abs(plain.start_ori() - measured) < 2e-6                # without the filter: the first kept point's azimuth
        # ring 0's first point: the first point of the ring-ordered cloud, whose azimuth the filter may adopt
        front = infer.cloud(0)[0]
        expect = model.update(measured, azimuth(float(front[0]), float(front[1])))
        assert abs(infer.start_ori() - expect) < 1e-5, (k, infer.start_ori(), expect)
        if k in stray and k >= 10:
            # the stray return is 2 rad away: the filter must not follow it
            assert abs(infer.start_ori() - measured) > 1.0
            replaced += 1
            # rel_time follows the inferred start: the stray point's own rel_time is no longer 0
            a, b = plain.cloud(0), infer.cloud(0)
            np.testing.assert_array_equal(a[:, :3], b[:, :3])
            assert np.abs((a[:, 3] - np.floor(a[:, 3])) - (b[:, 3] - np.floor(b[:, 3]))).max() > 0.01
        elif k < 10:
            np.testing.assert_array_equal(plain.cloud(0), infer.cloud(0))   # the history is not full yet: nothing changes
    assert replaced == 3


def test_filter_is_off_by_default_and_ignored_by_the_ring_field_overload(oracle):
    from pp_util import ring_field
    ds = synth.make_dataset("indoor", 1, 0.1)
    lid = ds.lidar
    sweeps = make_sweeps(ds.frames[0].scan, 13, 0.03, stray_at=(12,))
    a = capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings)
    b = capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings, _cfg(oracle, True))
    for scan in sweeps:
        ring = ring_field(scan, lid)
        a.process(scan, ring=ring)
        b.process(scan, ring=ring)
        np.testing.assert_array_equal(a.cloud(0), b.cloud(0))
