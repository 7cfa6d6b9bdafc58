"""CPU checks of the PointIR overload of PointToRing as the oracle restates it (PointProcessor.cc:428-536; the
reference constructs it with uneven = true for sensor_type 320, processor_node.cc:73)."""
import numpy as np

from lio_amd import capi, synth
from pp_util import ring_field


def _scan():
    ds = synth.make_dataset("indoor", 1, 0.1)
    return ds.lidar, ds.frames[0].scan


def test_ring_field_reproduces_the_binning_of_the_elevation_formula(oracle):
    lidar, scan = _scan()
    ring = ring_field(scan, lidar)
    even = capi.PointProcessor(oracle, lidar.lower_deg, lidar.upper_deg, lidar.rings)
    ir = capi.PointProcessor(oracle, lidar.lower_deg, lidar.upper_deg, lidar.rings)
    even.process(scan)
    ir.process(scan, ring=ring)
    np.testing.assert_array_equal(even.ring_offsets(), ir.ring_offsets())
    a, b = even.cloud(0), ir.cloud(0)
    np.testing.assert_array_equal(a[:, :3], b[:, :3])
    np.testing.assert_array_equal(np.floor(a[:, 3]), np.floor(b[:, 3]))       # integer part = ring
    for which in (1, 2, 3):                                                    # picks depend on xyz only
        np.testing.assert_array_equal(even.indices(which)[0], ir.indices(which)[0])
        np.testing.assert_array_equal(even.indices(which)[1], ir.indices(which)[1])
    # rel_time over the swept range instead of 2 pi: the last azimuth maps to exactly scan_period
    frac = b[:, 3] - np.floor(b[:, 3])
    assert frac.min() == 0.0 and abs(frac.max() - 0.1) < 2e-6
    assert (a[:, 3] - np.floor(a[:, 3])).max() < 0.1 - 1e-5                    # the 2 pi form never reaches it


def test_rel_time_formula(oracle):
    """rel_time = scan_period * (unwrapped azimuth - start_ori) / (end_ori - start_ori), float/double mix of :464-521."""
    lidar, scan = _scan()
    ring = ring_field(scan, lidar)
    pp = capi.PointProcessor(oracle, lidar.lower_deg, lidar.upper_deg, lidar.rings)
    pp.process(scan, ring=ring)
    out = pp.cloud(0)
    ok = np.isfinite(scan[:, :3]).all(axis=1)
    pts = scan[ok]
    azi = (2 * np.pi - np.arctan2(pts[:, 1], pts[:, 0]).astype(np.float32).astype(np.float64)).astype(np.float32)
    azi = np.where(azi >= 2 * np.pi, (azi - 2 * np.pi).astype(np.float32), azi)
    start = azi[0]
    unwrapped = np.where((azi - start) < 0, (azi.astype(np.float64) + 2 * np.pi).astype(np.float32), azi)
    end = max(np.float32(0), unwrapped.max())
    rel = (0.1 * (unwrapped - start).astype(np.float64) / np.float64(np.float32(end - start))).astype(np.float32)
    # bring the expected values into ring order (stable)
    order = np.argsort(ring[ok], kind="stable")
    expect = ring[ok][order].astype(np.float32) + rel[order]
    np.testing.assert_allclose(out[:, 3], expect, rtol=0, atol=4e-6)   # numpy's arctan2 vs libm atan2f: last-ulp differences
    np.testing.assert_array_equal(out[:, :3], pts[order][:, :3])


def test_ring_labels_are_only_labels_and_out_of_range_rings_are_dropped(oracle):
    lidar, scan = _scan()
    ring = ring_field(scan, lidar)
    R = lidar.rings
    fwd = capi.PointProcessor(oracle, -1, 1, R)   # the elevation bounds play no role in this overload
    rev = capi.PointProcessor(oracle, -1, 1, R)
    fwd.process(scan, ring=ring)
    rev.process(scan, ring=(R - 1 - ring).astype(np.uint16))
    of, orv = fwd.ring_offsets(), rev.ring_offsets()
    np.testing.assert_array_equal(np.diff(of), np.diff(orv)[::-1])
    cf, cr = fwd.cloud(0), rev.cloud(0)
    for r in range(R):
        np.testing.assert_array_equal(cf[of[r]:of[r + 1], :3], cr[orv[R - 1 - r]:orv[R - r], :3])
    # rings >= num_rings are skipped (:472-476); a processor built for fewer rings sees only the lower ones
    half = capi.PointProcessor(oracle, -1, 1, R // 2)
    half.process(scan, ring=ring)
    oh = half.ring_offsets()
    np.testing.assert_array_equal(oh, of[: R // 2 + 1])
    np.testing.assert_array_equal(half.cloud(0)[:, :3], cf[: of[R // 2], :3])
    # empty / all-NaN input
    for s in (np.zeros((0, 4), np.float32), np.full((20, 4), np.nan, np.float32)):
        e = capi.PointProcessor(oracle, -1, 1, R)
        e.process(s, ring=np.zeros(len(s), np.uint16))
        assert all(e.cloud(w).shape[0] == 0 for w in range(5))
