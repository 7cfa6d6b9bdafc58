#!/usr/bin/env python
"""PointProcessor alone on HDL-64E sweeps (for LIO_DEBUG_TIMING=1 and rocprofv3 runs)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
import numpy as np  # noqa: E402

from lio_amd import capi, synth  # noqa: E402

hip = capi.LioLib(capi.HIP_LIB_PATH)
ds = synth.make_dataset("outdoor", 4, 0.1)
pp = capi.PointProcessor(hip, ds.lidar.lower_deg, ds.lidar.upper_deg, ds.lidar.rings)
ms = []
for rep in range(5):
    for f in ds.frames:
        t = time.perf_counter()
        pp.process(f.scan)
        ms.append((time.perf_counter() - t) * 1e3)
print("median ms", float(np.median(ms[4:])), "points", ds.frames[0].scan.shape[0])
