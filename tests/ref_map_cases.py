"""The frame sequences on which the reference's own PointMapping was run for tests/golden/ref_mapping_digests.json — shared by the
generator (tests/golden/make_ref_mapping_digests.py, build container only) and tests/test_ref_mapping_digests.py.  Inputs are
regenerated (seeded); only transforms (bit patterns), the cube-window state and digests of the clouds are committed."""
import numpy as np

from mapping_util import drifting_inputs
from ref_odom_cases import bits
from ref_pp_cases import digest


def cases(oracle):
    """-> list of (name, [(corner_last, surf_last, (q_xyzw, p), set_init_flag_before), ...])"""
    out = []
    for kind, n in (("indoor", 5), ("outdoor", 3)):
        out.append((f"{kind}_sequence", [(c, s, T, False) for c, s, T, _ in drifting_inputs(oracle, kind, n)]))
    # the 21 x 21 x 11 window of 50 m cubes shifted back and forth along x (PointMapping.cc:808-989)
    rng = np.random.default_rng(1)
    pts = np.zeros((4000, 4), np.float32)
    pts[:, :3] = rng.uniform(-60, 60, (4000, 3))
    pts[:, 3] = rng.uniform(0, 16, 4000)
    out.append(("window_shift", [(pts[:500], pts, ([0, 0, 0, 1], [x, 30.0, -10.0]), False) for x in (0.0, 400.0, 380.0, -420.0, 1000.0)]))
    # SetInitFlag(true) before the third frame: no odometry increment, no map update (PointMapping.cc:781-783, 1021)
    fr = drifting_inputs(oracle, "indoor", 3)
    out.append(("frozen_after_imu_init", [(c, s, T, k == 2) for k, (c, s, T, _) in enumerate(fr)]))
    return out


def row_of(tobe, aft, clouds, center, valid, cubes):
    return {"tobe": bits(tobe), "aft": bits(aft), "clouds": [digest(c) for c in clouds], "center": [int(v) for v in center],
            "valid": [int(v) for v in valid], "cubes": digest(np.concatenate(cubes, 0) if cubes else np.zeros((0, 4), np.float32))}


def replay_lib(lib, frames):
    """the sequence through `lib`'s lio_map_* entry points"""
    from lio_amd import capi

    m = capi.PointMapping(lib)
    rows = []
    for corner, surf, T_sum, freeze in frames:
        if freeze:
            m.set_init_flag(True)
        r = m.process(corner, surf, T_sum)
        q, p = m.transform_tobe_mapped()
        cen, valid = m.cube_state()
        cubes = [m.cube(cls, i) for i in valid for cls in (0, 1)]
        rows.append(row_of(np.concatenate([q, p]), np.concatenate([r["T_aft"][0], r["T_aft"][1]]), [m.cloud(w) for w in range(4)], cen, valid, cubes))
    return rows
