#!/usr/bin/env python
"""Generates tests/golden/ref_degenerate_odometry.json: transform_es_ of THE REFERENCE'S OWN PointOdometry.cc (oracle/_ref/libref_odometry.so,
see oracle/ref_odometry.cc) on the degenerate scenes of tests/degenerate_util.py — a ground plane with 0 / 1 / 2 poles, which straddle
the scan-to-scan eigenvalue threshold of 10 (PointOdometry.cc:584-615).  Build container only."""
import ctypes as C
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
from lio_amd import capi  # noqa: E402
import degenerate_util as D  # noqa: E402
from ref_odom_cases import bits  # noqa: E402

spec = importlib.util.spec_from_file_location("odo", os.path.join(HERE, "make_ref_odometry_digests.py"))
odo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(odo)          # (the ctypes declarations of libref_odometry.so)


def main():
    ref, fp = odo.ref, odo.fp
    oracle = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    out = {}
    for name in D.ODOMETRY_SCENES:
        cls, _ = D.odometry_sweeps(oracle, name, 3)
        h = ref.ref_odom_create(0.1, 2, 25, 0)
        rows = []
        for k, cl in enumerate(cls):
            cl = [np.ascontiguousarray(c, np.float32) for c in cl]
            args = []
            for c in cl + [np.zeros((0, 4), np.float32)]:
                args += [c.ctypes.data_as(fp), len(c)]
            ref.ref_odom_process(h, *args, 1.0 + 0.1 * k)
            Te, Ts, fc = np.zeros(7, np.float32), np.zeros(7, np.float32), C.c_long(0)
            ref.ref_odom_get(h, Te.ctypes.data_as(fp), Ts.ctypes.data_as(fp), C.byref(fc))
            rows.append(bits(Te))
        ref.ref_odom_destroy(h)
        out[name] = rows
    path = os.path.join(HERE, "ref_degenerate_odometry.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print(path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
