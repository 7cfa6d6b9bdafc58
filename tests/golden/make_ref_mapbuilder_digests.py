#!/usr/bin/env python
"""Generates tests/golden/ref_mapbuilder_digests.json from THE REFERENCE'S OWN MapBuilder: src/map_builder/MapBuilder.cc over
src/point_processor/PointMapping.cc, compiled where they lie against the stand-in headers of oracle/ref_shim (`make -C oracle ref` ->
oracle/_ref/libref_mapbuilder.so).  Every frame goes in as the four messages of the odometry node (corner, surf, full cloud,
/laser_odom_to_init) through the reference's own handlers, then ProcessMap().  Per frame: transform_tobe_mapped_ /
transform_aft_mapped_ bit patterns, digests of the down-sampled stacks and the from-map clouds, the cube-window centre, the valid-cube
list and a digest of those cubes' contents.  Runs only in the build container; the output is committed."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
from lio_amd import capi  # noqa: E402
from ref_map_cases import row_of  # noqa: E402
from ref_mb_cases import CASES, frames_of  # noqa: E402

ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_mapbuilder.so"))
fp = C.POINTER(C.c_float)
ref.ref_mb_create.restype = C.c_void_p
ref.ref_mb_create.argtypes = [fp, C.c_int, C.c_int]
ref.ref_mb_destroy.argtypes = [C.c_void_p]
ref.ref_mb_process.argtypes = [C.c_void_p, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_size_t, fp, C.c_double]
ref.ref_mb_get_transform.argtypes = [C.c_void_p, C.c_int, fp]
ref.ref_mb_count.restype = C.c_size_t
ref.ref_mb_count.argtypes = [C.c_void_p, C.c_int]
ref.ref_mb_get_cloud.argtypes = [C.c_void_p, C.c_int, fp]
ref.ref_mb_cube_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.c_int]
ref.ref_mb_cube_count.restype = C.c_size_t
ref.ref_mb_cube_count.argtypes = [C.c_void_p, C.c_int, C.c_longlong]
ref.ref_mb_get_cube.argtypes = [C.c_void_p, C.c_int, C.c_longlong, fp]


def cloud(h, w):
    n = ref.ref_mb_count(h, w)
    a = np.zeros((n, 4), np.float32)
    if n:
        ref.ref_mb_get_cloud(h, w, a.ctypes.data_as(fp))
    return a


def cube(h, cls, idx):
    n = ref.ref_mb_cube_count(h, cls, int(idx))
    a = np.zeros((n, 4), np.float32)
    if n:
        ref.ref_mb_get_cube(h, cls, int(idx), a.ctypes.data_as(fp))
    return a


def main():
    oracle = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    dflt = capi.PointMapping(oracle).cfg
    out = {}
    for name, (_, _, e4, skip) in CASES.items():
        cfg = np.array([dflt.corner_filter_size, dflt.surf_filter_size, 0.6, dflt.min_match_sq_dis, dflt.min_plane_dis], np.float32)
        h = ref.ref_mb_create(cfg.ctypes.data_as(fp), e4, skip)
        rows = []
        for k, (corner, surf, T_sum) in enumerate(frames_of(oracle, name)):
            c, s = np.ascontiguousarray(corner, np.float32), np.ascontiguousarray(surf, np.float32)
            full = np.zeros((1, 4), np.float32)
            T7 = np.concatenate([np.asarray(T_sum[0], np.float32), np.asarray(T_sum[1], np.float32)])
            ref.ref_mb_process(h, c.ctypes.data_as(fp), len(c), s.ctypes.data_as(fp), len(s), full.ctypes.data_as(fp), 0, T7.ctypes.data_as(fp), 1.0 + 0.1 * k)
            tobe, aft = np.zeros(7, np.float32), np.zeros(7, np.float32)
            ref.ref_mb_get_transform(h, 0, tobe.ctypes.data_as(fp))
            ref.ref_mb_get_transform(h, 1, aft.ctypes.data_as(fp))
            cen, vi = (C.c_int * 3)(), (C.c_longlong * 256)()
            nv = ref.ref_mb_cube_state(h, cen, vi, 256)
            valid = list(vi[:nv])
            rows.append(row_of(tobe, aft, [cloud(h, w) for w in range(4)], list(cen), valid, [cube(h, cls, i) for i in valid for cls in (0, 1)]))
        ref.ref_mb_destroy(h)
        out[name] = rows
    path = os.path.join(HERE, "ref_mapbuilder_digests.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print(path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
