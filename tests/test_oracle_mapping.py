"""CPU checks of the oracle's scan-to-map stage (PointMapping.cc:325-1208 restated in oracle/mapping.h): it pulls a
drifting odometry pose back onto the map, the cube map obeys the VoxelGrid / window-shift semantics of the
reference, and UpdateMapDatabase re-bases the valid cubes on the current centre."""
import numpy as np

from lio_amd import capi
from mapping_util import drifting_inputs


def test_oracle_mapping_recovers_pose(oracle):
    frames = drifting_inputs(oracle, "indoor", 5)
    mp = capi.PointMapping(oracle)
    for k, (corner, surf, T_sum, p_gt) in enumerate(frames):
        r = mp.process(corner, surf, T_sum)
        if k == 0:
            # empty map: OptimizeTransformTobeMapped returns before TransformUpdate (PointMapping.cc:327-329)
            assert r["iterations"] == 0
            assert np.allclose(r["T_aft"][1], 0) and np.allclose(r["T_aft"][0], [0, 0, 0, 1])
            continue
        assert r["num_selected"] > 3000
        assert np.linalg.norm(r["T_aft"][1][:2] - p_gt[:2]) < 0.03   # the input was off by 0.06 m * k
        assert abs(r["T_aft"][1][2] - p_gt[2]) < 0.08                # 16 rings: weak vertical constraint
        assert np.linalg.norm(T_sum[1] - p_gt) > 0.06 * k * 0.99
    cen, valid = mp.cube_state()
    assert cen == [10, 10, 5] and 100 < len(valid) <= 125
    # from_map of the last call = concatenation of the valid cubes as they were BEFORE its own update
    score, point, coeff = mp.score_point_coeff()
    assert len(score) >= 50 and np.all(np.diff(score) <= 0)
    assert np.allclose(np.linalg.norm(coeff[:, :3], axis=1), 1.0, atol=1e-5)  # abs_coeff: unit normal + offset


def test_oracle_cube_map_is_voxel_filtered_and_idempotent(oracle):
    frames = drifting_inputs(oracle, "indoor", 3)
    mp = capi.PointMapping(oracle)
    for corner, surf, T_sum, _ in frames:
        mp.process(corner, surf, T_sum)
    cen, valid = mp.cube_state()
    total = 0
    for idx in valid:
        for cls, leaf in ((0, 0.2), (1, 0.4)):
            c = mp.cube(cls, idx)
            total += len(c)
            if len(c) > 1:
                # one point per voxel of the cube's own grid
                mn = np.floor(c[:, :3].min(axis=0) / np.float32(leaf))
                v = np.floor(c[:, :3] * (np.float32(1.0) / np.float32(leaf))) - mn
                assert len(np.unique(v, axis=0)) == len(c)
    assert total > 5000
    before = {(cls, int(i)): mp.cube(cls, i) for i in valid for cls in (0, 1)}
    # re-filtering filtered cubes with nothing new is the identity
    mp.update_map_database(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), valid, ([0, 0, 0, 1], [0, 0, 0]), cen)
    for (cls, i), c in before.items():
        np.testing.assert_array_equal(mp.cube(cls, i), c)


def test_oracle_window_shift_drops_far_cubes(oracle):
    mp = capi.PointMapping(oracle)
    rng = np.random.default_rng(1)
    pts = np.zeros((500, 4), np.float32)
    pts[:, :3] = rng.uniform(-20, 20, (500, 3))
    mp.process(pts[:100], pts, ([0, 0, 0, 1], [0, 0, 0]))
    cen, _ = mp.cube_state()
    home = 10 + 21 * 10 + 441 * 5
    assert cen == [10, 10, 5] and len(mp.cube(1, home)) > 0
    # 400 m in +x: the sensor cube would be 10+8=18 >= 21-3, the window shifts by one cube (PointMapping.cc:838-853)
    mp.process(pts[:100], pts, ([0, 0, 0, 1], [400, 0, 0]))
    cen, _ = mp.cube_state()
    assert cen == [9, 10, 5]
    assert len(mp.cube(1, home - 1)) > 0 and len(mp.cube(1, home)) == 0   # same cube, index moved by -1
    # 1000 m: far beyond the window; every old cube falls off the low side
    mp.process(pts[:100], pts, ([0, 0, 0, 1], [1000, 0, 0]))
    cen, _ = mp.cube_state()
    assert cen[0] < 0 and sum(len(mp.cube(1, i)) for i in range(21 * 21 * 11)) == len(mp.cube(1, 17 + 21 * 10 + 441 * 5))


def test_oracle_map_builder_4dof(oracle):
    """MapBuilder::ProcessMap (MapBuilder.cc:220-558): the first call adopts the odometry pose, every skip_count-th
    call optimises (4-DoF weighting), the others only commit the prediction; the map grows on every call."""
    frames = drifting_inputs(oracle, "indoor", 5)
    mb = capi.PointMapping(oracle, map_builder=1, enable_4d=1, skip_count=2)
    its = []
    for k, (corner, surf, T_sum, p_gt) in enumerate(frames):
        r = mb.process(corner, surf, T_sum)
        its.append(r["iterations"])
        if k == 0:
            np.testing.assert_allclose(mb.transform_tobe_mapped()[1], T_sum[1], atol=1e-6)
            np.testing.assert_allclose(np.abs(mb.transform_tobe_mapped()[0]), np.abs(np.asarray(T_sum[0], np.float32)), atol=1e-6)
    # call 0: map empty -> OptimizeMap returns early; calls 2, 4 optimise; calls 1, 3 are skipped
    assert its[0] == 0 and its[1] == 0 and its[3] == 0 and its[2] > 0 and its[4] > 0
    q, p = mb.transform_tobe_mapped()
    assert np.linalg.norm(p[:2] - frames[-1][3][:2]) < 0.08      # pulled back from 0.23 m of injected drift
    assert len(mb.score_point_coeff()[0]) == 0                   # OptimizeMap keeps no score list
    _, valid = mb.cube_state()
    assert sum(len(mb.cube(1, i)) for i in valid) > 10000
