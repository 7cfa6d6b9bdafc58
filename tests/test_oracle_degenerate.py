"""CPU checks of the degeneracy scenes (tests/degenerate_util.py) on the oracle: the scene families really sit on both
sides of the thresholds, kz is reported through every entry point, the masked update components are zero, and the
per-iteration record of the scan-to-scan step is consistent with its final transform."""
import numpy as np
import pytest

from lio_amd import capi
from degenerate_util import (ESTIMATOR_SCENES, ODOMETRY_SCENES, estimator_pair, masked_rotation_components, odometry_sweeps)
from mapping_util import drifting_inputs
import degenerate_util as du


@pytest.mark.parametrize("scene", list(ESTIMATOR_SCENES))
def test_estimator_newest_frame_kz(oracle, scene):
    _, (est,), kz, _ = estimator_pair((oracle,), scene)
    rep = est.solve()
    assert rep.laser_odom_kz == kz
    assert 1 <= rep.laser_odom_iterations <= 10


def test_odometry_scenes_straddle_the_threshold_and_mask_rotation(oracle):
    kzs = {}
    for scene in ODOMETRY_SCENES:
        cl, _ = odometry_sweeps(oracle, scene, 2)
        od = capi.PointOdometry(oracle, 0.1, 2, 25, False)
        od.process(*cl[0])
        r = od.process(*cl[1])
        kzs[scene] = r["kz"]
        assert r["trace"].shape == (r["iterations"], 7)
        np.testing.assert_array_equal(r["trace"][-1][4:], r["T_es"][1])          # the last record is the final translation
        masked, free = masked_rotation_components(r["trace"], [0, 0, 0, 1], r["kz"])
        assert masked < 1e-7                                                      # zeroed components: no rotation about those axes
        if r["kz"] < 3:
            assert free > 1e-6
    assert kzs["ground"] == 3 and kzs["ground_one_pole"] == 1 and kzs["ground_two_poles"] == 0, kzs


def test_mapping_and_keyframe_kz(oracle):
    from kf_util import keyframe_inputs, load
    got = {}
    for scene, (factory, sigma, _) in du.MAPPING_SCENES.items():
        from lio_amd import synth
        frames = drifting_inputs(oracle, "indoor", 2, scene=factory(), traj=synth.traj_corridor(), range_sigma=sigma)
        mp = capi.PointMapping(oracle)
        mp.process(*frames[0][:3])
        r = mp.process(*frames[1][:3])
        got[scene] = (r["degenerate"], r["kz"])
    assert got["ground"] == (1, 3) and got["corridor_below_threshold"] == (1, 1) and got["corridor_above_threshold"] == (0, 0), got
    # MapBuilder's 4-DoF weights (roll / pitch columns x 5e-3, MapBuilder.cc:903-914) put two eigenvalues below 100 in ANY scene
    maps, kfs = keyframe_inputs(oracle, "indoor", 2, 2, map_builder=1)
    b = load(capi.KeyframeBatch(oracle, map_builder=1, enable_4d=1), maps, kfs)
    r = b.refine()
    assert np.all(r["kz"] == 2), r["kz"]
    b6 = load(capi.KeyframeBatch(oracle), maps, kfs)
    assert np.all(b6.refine()["kz"] == 0)
