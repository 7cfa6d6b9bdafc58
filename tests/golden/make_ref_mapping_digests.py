#!/usr/bin/env python
"""Generates tests/golden/ref_mapping_digests.json from THE REFERENCE'S OWN PointMapping: src/point_processor/PointMapping.cc
compiled where it lies against the stand-in headers of oracle/ref_shim (`make -C oracle ref` -> oracle/_ref/libref_mapping.so;
pcl::VoxelGrid forwards to the oracle's restatement, the kd-tree is an exact search, Eigen's ColPivHouseholderQR /
SelfAdjointEigenSolver forward to the oracle's restatements).  Every frame goes in as a /compact_data message through the
reference's own CompactDataHandler, then Process().  Per frame: transform_tobe_mapped_ / transform_aft_mapped_ bit patterns, digests
of the down-sampled stacks and the from-map clouds, the cube-window centre, the valid-cube list and a digest of those cubes'
contents.  Runs only in the build container; the output is committed."""
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
from ref_map_cases import cases, row_of  # noqa: E402

ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_mapping.so"))
fp = C.POINTER(C.c_float)
ref.ref_map_create.restype = C.c_void_p
ref.ref_map_create.argtypes = [C.c_float, C.c_int]
ref.ref_map_destroy.argtypes = [C.c_void_p]
ref.ref_map_set_init_flag.argtypes = [C.c_void_p, C.c_int]
ref.ref_map_process_compact.argtypes = [C.c_void_p, fp, C.c_size_t, C.c_double]
ref.ref_map_get_transform.argtypes = [C.c_void_p, C.c_int, fp]
ref.ref_map_count.restype = C.c_size_t
ref.ref_map_count.argtypes = [C.c_void_p, C.c_int]
ref.ref_map_get_cloud.argtypes = [C.c_void_p, C.c_int, fp]
ref.ref_map_cube_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.c_int]
ref.ref_map_cube_count.restype = C.c_size_t
ref.ref_map_cube_count.argtypes = [C.c_void_p, C.c_int, C.c_longlong]
ref.ref_map_get_cube.argtypes = [C.c_void_p, C.c_int, C.c_longlong, fp]


def cloud(h, w):
    n = ref.ref_map_count(h, w)
    a = np.zeros((n, 4), np.float32)
    if n:
        ref.ref_map_get_cloud(h, w, a.ctypes.data_as(fp))
    return a


def cube(h, cls, idx):
    n = ref.ref_map_cube_count(h, cls, int(idx))
    a = np.zeros((n, 4), np.float32)
    if n:
        ref.ref_map_get_cube(h, cls, int(idx), a.ctypes.data_as(fp))
    return a


def main():
    oracle = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    out = {}
    for name, frames in cases(oracle):
        h = ref.ref_map_create(0.1, 10)
        rows = []
        for k, (corner, surf, T_sum, freeze) in enumerate(frames):
            if freeze:
                ref.ref_map_set_init_flag(h, 1)
            # the wire message the reference's odometry node would send (lio_compact_encode is pinned to PointOdometry.cc's packing)
            msg = np.ascontiguousarray(oracle.compact_encode(capi.TransformF.make(*T_sum), corner, surf, np.zeros((0, 4), np.float32)), np.float32)
            ref.ref_map_process_compact(h, msg.ctypes.data_as(fp), len(msg), 1.0 + 0.1 * k)
            tobe, aft = np.zeros(7, np.float32), np.zeros(7, np.float32)
            ref.ref_map_get_transform(h, 0, tobe.ctypes.data_as(fp))
            ref.ref_map_get_transform(h, 1, aft.ctypes.data_as(fp))
            cen, vi = (C.c_int * 3)(), (C.c_longlong * 256)()
            nv = ref.ref_map_cube_state(h, cen, vi, 256)
            valid = list(vi[:nv])
            rows.append(row_of(tobe, aft, [cloud(h, w) for w in range(4)], list(cen), valid, [cube(h, cls, i) for i in valid for cls in (0, 1)]))
        ref.ref_map_destroy(h)
        out[name] = rows
    path = os.path.join(HERE, "ref_mapping_digests.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print(path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
