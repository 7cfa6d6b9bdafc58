"""The frame sequences on which the reference's own MapBuilder was run for tests/golden/ref_mapbuilder_digests.json — shared by the
generator (tests/golden/make_ref_mapbuilder_digests.py, build container only) and tests/test_ref_mapbuilder_digests.py.  Inputs are
regenerated (seeded); only transforms (bit patterns), the cube-window state and digests of the clouds are committed."""
from mapping_util import drifting_inputs

# name -> (kind, frames, enable_4d, skip_count)
CASES = {
    "indoor_4d": ("indoor", 6, 1, 2),      # OptimizeMap on even frames, Transform4DUpdate on odd ones (MapBuilder.cc:529-544)
    "outdoor_4d": ("outdoor", 4, 1, 2),
    "indoor_4d_every_frame": ("indoor", 4, 1, 1),
    "indoor_6d": ("indoor", 4, 0, 2),      # enable_4d off: OptimizeTransformTobeMapped / TransformUpdate behind the same gate
}


def frames_of(oracle, name):
    kind, n, _, _ = CASES[name]
    return [(c, s, T) for c, s, T, _ in drifting_inputs(oracle, kind, n)]


def replay_lib(lib, name, frames):
    """the sequence through `lib`'s lio_map_* entry points in MapBuilder mode"""
    import numpy as np

    from lio_amd import capi
    from ref_map_cases import row_of

    _, _, e4, skip = CASES[name]
    m = capi.PointMapping(lib, map_builder=1, enable_4d=e4, skip_count=skip)
    rows = []
    for corner, surf, T_sum in frames:
        r = m.process(corner, surf, T_sum)
        q, p = m.transform_tobe_mapped()
        cen, valid = m.cube_state()
        cubes = [m.cube(cls, i) for i in valid for cls in (0, 1)]
        rows.append(row_of(np.concatenate([q, p]), np.concatenate([r["T_aft"][0], r["T_aft"][1]]), [m.cloud(w) for w in range(4)], cen, valid, cubes))
    return rows
