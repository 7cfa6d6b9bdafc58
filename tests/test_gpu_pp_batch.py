"""lio_pp_process_batch over handles of ONE sensor: a single launch chain for all sweeps (every kernel of pointproc.hip once, the sweep
in blockIdx.z; PointProcessor.cc:207-783) — against the same sweeps one call at a time (bit for bit: one sweep is the B = 1 case of the
same kernels), against the oracle, and against the digests of the reference's own PointProcessor.cc
(tests/golden/ref_pointproc_digests.json)."""
import numpy as np
import pytest
import torch  # noqa: F401  (before the product library is loaded: both bring a HIP runtime, torch must come first)

from lio_amd import capi, synth
from ref_pp_cases import cases, digest
from test_gpu_parity import _assert_rel_time_close
from test_ref_pointproc_digests import CLOUDS, GOLD, ORDER

pytestmark = pytest.mark.gpu


def _same(a, b, start_ori=True):
    for which in range(5):
        np.testing.assert_array_equal(a.cloud(which), b.cloud(which))
    for which in (1, 2, 3):
        ra, ia = a.indices(which)
        rb, ib = b.indices(which)
        np.testing.assert_array_equal(ra, rb)
        np.testing.assert_array_equal(ia, ib)
    np.testing.assert_array_equal(a.ring_offsets(), b.ring_offsets())
    np.testing.assert_array_equal(a.ring_intensity(), b.ring_intensity())
    ca, ma = a.curvature()
    cb, mb = b.curvature()
    np.testing.assert_array_equal(ca, cb)
    np.testing.assert_array_equal(ma, mb)
    if start_ori:
        sa, sb = a.start_ori(), b.start_ori()
        assert sa == sb or (np.isnan(sa) and np.isnan(sb))


def _pp(lib, lid, **over):
    cfg = capi.PPConfig()
    lib.dll.lio_pp_default_config(cfg)
    for k, v in over.items():
        setattr(cfg, k, v)
    return capi.PointProcessor(lib, lid.lower_deg, lid.upper_deg, lid.rings, cfg)


@pytest.fixture(scope="module")
def hdl_sweeps():
    ds = synth.make_dataset("outdoor", 4, 0.1)
    sweeps = [f.scan for f in ds.frames]
    sweeps.append(sweeps[0][: sweeps[0].shape[0] * 2 // 3].copy())          # ragged: a sweep cut short
    sweeps.append(sweeps[1][5000:].copy())                                  # ... and one that starts elsewhere
    return ds.lidar, sweeps


def test_one_chain_equals_one_by_one(hip, hdl_sweeps):
    """Six HDL-64E sweeps of different lengths through six handles of one sensor: clouds, pick lists, ring offsets, curvature, masks,
    intensities and start azimuths equal the single calls bit for bit — three calls, the batch shrinking and growing (every buffer and
    the shared processor are reused), an empty sweep in the middle of the last one."""
    lid, sweeps = hdl_sweeps
    B = len(sweeps)
    batch = [_pp(hip, lid) for _ in range(B)]
    single = [_pp(hip, lid) for _ in range(B)]
    empty = np.zeros((0, 4), np.float32)
    plans = [list(range(B)), [3, 0, 5], [1, 4, None, 2, 0, 3]]
    for plan in plans:
        ins = [sweeps[k] if k is not None else empty for k in plan]
        capi.PointProcessor.process_batch(batch[: len(plan)], ins)
        for h, x in zip(single, ins):
            h.process(x)
        for a, b, x in zip(batch, single, ins):
            _same(a, b, start_ori=x.shape[0] > 0)     # (an empty sweep alone keeps the handle's last start azimuth, as the reference's member does; in a batch it reads NaN)
            assert a.cloud(0).shape[0] == (0 if x.shape[0] == 0 else b.cloud(0).shape[0])
    # a later batch that reuses the shared storage without one of the earlier handles: that handle is told so (no other sweep's data)
    capi.PointProcessor.process_batch([batch[0], batch[2]], [sweeps[0], sweeps[1]])
    assert batch[3].cloud(0).shape[0] == 0
    with pytest.raises(capi.LioError):
        batch[3].ring_offsets()
    batch[3].process(sweeps[3])
    single[3].process(sweeps[3])
    _same(batch[3], single[3])
    single[0].process(sweeps[0]); single[2].process(sweeps[1])
    _same(batch[0], single[0]); _same(batch[2], single[2])
    # a handle of the batch used on its own again answers for its own sweep
    batch[1].process(sweeps[2])
    single[1].process(sweeps[2])
    _same(batch[1], single[1])
    _same(batch[0], single[0])            # ... while its neighbours still hold the batch's results


def test_one_chain_from_device_memory(hip, hdl_sweeps):
    """lio_pp_process_batch_device: the same sweeps resident in HBM (torch tensors), no transfer over PCIe."""
    lid, sweeps = hdl_sweeps
    dev = [torch.from_numpy(np.ascontiguousarray(s, np.float32)).cuda() for s in sweeps]
    torch.cuda.synchronize()
    B = len(sweeps)
    batch = [_pp(hip, lid) for _ in range(B)]
    single = [_pp(hip, lid) for _ in range(B)]
    for rep in range(2):
        capi.PointProcessor.process_batch_device(batch, [t.data_ptr() for t in dev], [t.shape[0] for t in dev])
        for h, x in zip(single, sweeps):
            h.process(x)
        for a, b in zip(batch, single):
            _same(a, b)
    one = _pp(hip, lid)                    # B = 1: the handle's own chain, input from device memory
    capi.PointProcessor.process_batch_device([one], [dev[2].data_ptr()], [dev[2].shape[0]])
    _same(one, single[2])


NO_RING = [c for c in cases() if all(r is None for _, r in c[3]) and c[0] != "indoor_infer_start_ori"]


@pytest.mark.parametrize("case", NO_RING, ids=[c[0] for c in NO_RING])
def test_one_chain_follows_the_reference(hip, oracle, case):
    """Every sweep of a reference case twice in one batch (so that the call takes the one-chain path whatever the case's sweep count):
    coordinates, order and counts equal the oracle's, whose digests are those of the reference's own PointProcessor.cc."""
    name, lid, over, sweeps = case
    over = {k: v for k, v in over.items() if k != "uneven"}
    scans = [s for s, _ in sweeps] * 2
    hs = [_pp(hip, lid, **over) for _ in scans]
    capi.PointProcessor.process_batch(hs, scans)
    orc = _pp(oracle, lid, **over)
    for j, scan in enumerate(scans):
        k = j % len(sweeps)
        orc.process(scan)
        for c, w in zip(CLOUDS, ORDER):
            a, b = hs[j].cloud(w), orc.cloud(w)
            assert digest(b) == GOLD[name][k][c]
            assert a.shape == b.shape
            np.testing.assert_array_equal(a[:, :3], b[:, :3])
            np.testing.assert_allclose(a[:, 3], b[:, 3], atol=8e-6)
        for which in (1, 2, 3):
            np.testing.assert_array_equal(hs[j].indices(which)[0], orc.indices(which)[0])
            np.testing.assert_array_equal(hs[j].indices(which)[1], orc.indices(which)[1])


RING = [c for c in cases() if any(r is not None for _, r in c[3])]


@pytest.mark.parametrize("case", RING, ids=[c[0] for c in RING])
def test_one_chain_of_ring_field_sweeps_follows_the_reference(hip, oracle, case):
    """lio_pp_process_rings_batch: the PointIR overload of PointToRing (PointProcessor.cc:428-536; the two ring-field cases of the reference
    digests) for three sweeps in one chain — the case's sweep, the same sweep cut short, the sweep again: equal to single calls bit for
    bit, and to the oracle / the reference's digests like the single-sweep test."""
    name, lid, over, sweeps = case
    over = {k: v for k, v in over.items() if k != "uneven"}
    scan, ring = sweeps[0]
    m = scan.shape[0] * 3 // 4
    ins = [(scan, ring), (scan[:m].copy(), ring[:m].copy()), (scan, ring)]
    hs = [_pp(hip, lid, **over) for _ in ins]
    single = [_pp(hip, lid, **over) for _ in ins]
    capi.PointProcessor.process_rings_batch(hs, [s for s, _ in ins], [r for _, r in ins])
    for h, (s, r) in zip(single, ins):
        h.process(s, r)
    for a, b in zip(hs, single):
        _same(a, b)
    orc = _pp(oracle, lid, **over)
    orc.process(scan, ring)
    for c, w in zip(CLOUDS, ORDER):
        a, b = hs[2].cloud(w), orc.cloud(w)
        assert digest(b) == GOLD[name][0][c]
        assert a.shape == b.shape
        np.testing.assert_array_equal(a[:, :3], b[:, :3])
        np.testing.assert_allclose(a[:, 3], b[:, 3], atol=8e-6)


def test_one_chain_keeps_every_handles_start_azimuth_history(hip, oracle):
    """infer_start_ori (PointProcessor.cc:348-387) in a batch: three sensors, each with its own ten-sweep history, 24 rounds of one batch
    call; sensor j sees the reference case's sweeps shifted by 5 j.  Equal to three handles fed one call at a time, bit for bit, and
    sensor 0 follows the reference's digests through the oracle."""
    name, lid, over, sweeps = [c for c in cases() if c[0] == "indoor_infer_start_ori"][0]
    scans = [s for s, _ in sweeps]
    n = len(scans)
    batch = [_pp(hip, lid, **over) for _ in range(3)]
    single = [_pp(hip, lid, **over) for _ in range(3)]
    orc = _pp(oracle, lid, **over)
    n_wrapped = 0
    for k in range(n):
        ins = [scans[(k + 5 * j) % n] for j in range(3)]
        capi.PointProcessor.process_batch(batch, ins)
        for h, x in zip(single, ins):
            h.process(x)
        for a, b in zip(batch, single):
            _same(a, b)
        orc.process(scans[k])
        assert abs(batch[0].start_ori() - orc.start_ori()) < 2e-6        # atan2f ulp
        for c, w in zip(CLOUDS, ORDER):
            a, b = batch[0].cloud(w), orc.cloud(w)
            assert digest(b) == GOLD[name][k][c]
            assert a.shape == b.shape
            np.testing.assert_array_equal(a[:, :3], b[:, :3])
            # (an inferred start azimuth lands anywhere: a point within atan2f rounding of it wraps by one scan period on one side)
            n_wrapped += _assert_rel_time_close(a, b, orc.start_ori())
    assert n_wrapped < n * 40
