import sys, time, numpy as np
sys.path.insert(0, "lio-mapping_amd"); sys.path.insert(0, ".")  # run from the repo root: python profiles/mapping_profile.py
import bench
from lio_amd import capi
hip = capi.load_hip()
ds = bench.make_dataset("outdoor", 6, 0.0)
clouds, _ = bench.feature_clouds(hip, ds)
for rep in range(3):
    t = time.perf_counter()
    st = bench.mapping_ms_per_scan(hip, ds, clouds, n_frames=10)
    print(st, "%.1f ms total" % ((time.perf_counter() - t) * 1e3))
