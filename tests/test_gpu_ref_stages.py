"""The PRODUCT's scan-to-scan odometry, scan-to-map stage and MapBuilder mode on the GPU against what the REFERENCE's own sources
produced (tests/golden/ref_{odometry,mapping,mapbuilder}_digests.json: PointOdometry.cc, PointMapping.cc and MapBuilder.cc compiled
where they lie, see oracle/ref_*.cc; the oracle equals those files bit for bit, tests/test_ref_*_digests.py).

Same sequences and the same bounds as the product-vs-oracle tests of these stages (tests/test_gpu_parity.py: transform_es_ 1e-5,
transform_sum_ 1e-4; tests/test_gpu_mapping.py: 1e-4 m / 1e-4 rad, equal window state, clouds up to counted voxel-face flips) — what changes is the
right-hand side: the transforms compared against are the bit patterns the reference's code wrote.  (fp32 sums in a different order
are why these are tolerances and not equal bits; the PointProcessor, which has no such sums, is held to the reference bit for bit by
tests/test_gpu_ref_pointproc.py.)"""
import json
import os

import numpy as np
import pytest

from lio_amd import capi, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gold(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


def _f(bits):
    return np.array(bits, np.uint32).view(np.float32).astype(float)


def _count(d):
    return int(d.split(":")[0])


def _pose_gap(a, b):
    """(max |dp|, max quaternion component gap up to sign) between two [q, p] 7-vectors"""
    return float(np.abs(a[4:] - b[4:]).max()), float(min(np.abs(a[:4] - b[:4]).max(), np.abs(a[:4] + b[:4]).max()))


@pytest.mark.parametrize("name", ["indoor_io2", "outdoor_io3"])
def test_point_odometry_matches_the_reference(hip, oracle, name):
    import ref_odom_cases as oc

    case = [c for c in oc.cases() if c[0] == name][0]
    _, kind, n, io, no_deskew, _ = case
    want = _gold("ref_odometry_digests.json")[name]
    sweeps, _, lid = synth.make_sweeps(kind, n)
    od = capi.PointOdometry(hip, 0.1, io, 25, bool(no_deskew))
    worst = [0.0, 0.0]
    for k, sw in enumerate(sweeps):
        r = od.process(*oc.feature_clouds(oracle, lid, sw))
        es, ts = np.concatenate([r["T_es"][0], r["T_es"][1]]).astype(float), np.concatenate([r["T_sum"][0], r["T_sum"][1]]).astype(float)
        ges, gts = _f(want[k]["T_es"]), _f(want[k]["T_sum"])
        worst = [max(worst[0], float(np.abs(es - ges).max())), max(worst[1], float(np.abs(ts - gts).max()))]
        if k > 0:                                             # (the first sweep only initialises the odometry)
            np.testing.assert_allclose(es, ges, atol=1e-5)
        np.testing.assert_allclose(ts[4:], gts[4:], atol=1e-4)
        assert min(np.abs(ts[:4] - gts[:4]).max(), np.abs(ts[:4] + gts[:4]).max()) < 1e-4
        assert abs(len(od.last_cloud(0)) - _count(want[k]["last_corner"])) <= 0 and abs(len(od.last_cloud(1)) - _count(want[k]["last_surf"])) <= 0
    print(name, "product vs PointOdometry.cc: worst |dT_es|", worst[0], "worst |dT_sum|", worst[1])


@pytest.mark.parametrize("name", ["indoor_sequence", "outdoor_sequence"])
def test_point_mapping_matches_the_reference(hip, oracle, name):
    """Transforms and the cube window against the reference's bit patterns / lists.  The clouds are committed as count:sha256 only, and
    a VoxelGrid that sits downstream of transform_tobe_mapped_ (matched to 1e-4, not to the bit) cannot reproduce a digest; so, as in
    tests/test_gpu_ref_pointproc.py: (1) the ORACLE run beside the product reproduces the reference's digest of every cloud bit for bit
    (the oracle's clouds ARE PointMapping.cc's), (2) the product's clouds are compared with those clouds point by point, voxel-face
    flips counted (tests/test_gpu_mapping._cloud_mismatch), bounded by max(4, 0.5 %) and printed."""
    import ref_map_cases as mc
    from ref_pp_cases import digest
    from test_gpu_mapping import _cloud_mismatch

    frames = dict(mc.cases(oracle))[name]
    want = _gold("ref_mapping_digests.json")[name]
    m, mo = capi.PointMapping(hip), capi.PointMapping(oracle)
    worst = [0.0, 0.0]
    flips = {}
    for k, (corner, surf, T_sum, _) in enumerate(frames):
        r = m.process(corner, surf, T_sum)
        mo.process(corner, surf, T_sum)
        q, p = m.transform_tobe_mapped()
        for got, key in ((np.concatenate([q, p]).astype(float), "tobe"), (np.concatenate([r["T_aft"][0], r["T_aft"][1]]).astype(float), "aft")):
            dp, dq = _pose_gap(got, _f(want[k][key]))
            worst = [max(worst[0], dp), max(worst[1], dq)]
            assert dp < 1e-4 and dq < 1e-4, (name, k, key, dp, dq)
        cen, valid = m.cube_state()
        assert [int(v) for v in cen] == want[k]["center"] and [int(v) for v in valid] == want[k]["valid"], (name, k)
        for w, atol in ((capi.PointMapping.CORNER_STACK_DS, 1e-4), (capi.PointMapping.SURF_STACK_DS, 1e-4),
                        (capi.PointMapping.CORNER_FROM_MAP, 5e-4), (capi.PointMapping.SURF_FROM_MAP, 5e-4)):
            ref_cloud = mo.cloud(w)
            assert digest(ref_cloud) == want[k]["clouds"][w], (name, k, w)        # the oracle's cloud is the reference's, bit for bit
            got_cloud = m.cloud(w)
            bad = _cloud_mismatch(got_cloud, ref_cloud, atol)
            flips[(k, w)] = (bad, len(got_cloud), len(ref_cloud))
            assert bad <= max(4, len(ref_cloud) // 200), (name, k, w, bad, got_cloud.shape, ref_cloud.shape)
    print(name, "product vs PointMapping.cc: worst |dp|", worst[0], "worst |dq|", worst[1])
    print(name, "clouds (frame, which) -> (points without a partner + size gap, product size, reference size):", flips)


def test_map_builder_matches_the_reference(hip, oracle):
    import ref_mb_cases as bc

    name = "indoor_4d"
    _, _, e4, skip = bc.CASES[name]
    want = _gold("ref_mapbuilder_digests.json")[name]
    m = capi.PointMapping(hip, map_builder=1, enable_4d=e4, skip_count=skip)
    worst = [0.0, 0.0]
    for k, (corner, surf, T_sum) in enumerate(bc.frames_of(oracle, name)[:5]):
        r = m.process(corner, surf, T_sum)
        q, p = m.transform_tobe_mapped()
        for got, key in ((np.concatenate([q, p]).astype(float), "tobe"), (np.concatenate([r["T_aft"][0], r["T_aft"][1]]).astype(float), "aft")):
            dp, dq = _pose_gap(got, _f(want[k][key]))
            worst = [max(worst[0], dp), max(worst[1], dq)]
            assert dp < 1e-4 and dq < 1e-4, (k, key, dp, dq)
    print("product vs MapBuilder.cc: worst |dp|", worst[0], "worst |dq|", worst[1])


@pytest.mark.parametrize("scene,kz", [("corridor_below_threshold", 1), ("corridor_above_threshold", 0)])
def test_scan_to_map_degeneracy_matches_the_reference(hip, oracle, scene, kz):
    """The corridor on either side of the eigenvalue threshold (PointMapping.cc:650-680), frame 1 as in tests/test_gpu_degenerate.py,
    against the transform the reference's PointMapping.cc wrote (tests/golden/ref_degenerate_mapping.json)."""
    import degenerate_util as du
    from mapping_util import drifting_inputs

    factory, sigma, _ = du.MAPPING_SCENES[scene]
    frames = drifting_inputs(oracle, "indoor", 2, scene=factory(), traj=synth.traj_corridor(), range_sigma=sigma)
    want = _gold("ref_degenerate_mapping.json")[scene]
    m = capi.PointMapping(hip)
    m.process(*frames[0][:3])
    r = m.process(*frames[1][:3])
    assert r["kz"] == kz
    dp, dq = _pose_gap(np.concatenate([r["T_aft"][0], r["T_aft"][1]]).astype(float), _f(want[1]))
    print(scene, "product vs PointMapping.cc: |dp|", dp, "|dq|", dq)
    assert dp < 1e-4 and dq < 1e-4


def test_scan_to_scan_on_the_two_pole_ground_matches_the_reference(hip, oracle):
    """The regular member of the ground-plane family (kz = 0, smallest eigenvalue just above the scan-to-scan threshold of 10), as in
    tests/test_gpu_degenerate.py, against the transform_es_ the reference's PointOdometry.cc wrote
    (tests/golden/ref_degenerate_odometry.json)."""
    import degenerate_util as du

    cl, singular = du.odometry_sweeps(oracle, "ground_two_poles", 3)
    assert not singular
    want = _gold("ref_degenerate_odometry.json")["ground_two_poles"]
    od = capi.PointOdometry(hip, 0.1, 2, 25, False)
    worst = 0.0
    for k, c in enumerate(cl):
        r = od.process(*c)
        if k == 0:
            continue
        assert r["kz"] == 0
        es = np.concatenate([r["T_es"][0], r["T_es"][1]]).astype(float)
        worst = max(worst, float(np.abs(es - _f(want[k])).max()))
        np.testing.assert_allclose(es, _f(want[k]), atol=1e-5)
    print("product vs PointOdometry.cc on the two-pole ground: worst |dT_es|", worst)
