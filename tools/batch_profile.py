#!/usr/bin/env python
"""B copies of the headline window through lio_est_batch, `steps` restore + solve steps: the command the batched-solve kernel
profiles (profiles/r5_batch*_kernel_stats.md) are taken of.  Usage: batch_profile.py B [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
import torch  # noqa: E402,F401  (torch before the product library: both bring a HIP runtime)

import bench  # noqa: E402
from lio_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hip = capi.load_hip()
kind, W, Wo = "outdoor", 15, 5
ds = bench.make_dataset(kind, W)
clouds, _ = bench.feature_clouds(hip, ds)
est0 = bench.make_estimator(hip, ds, clouds, kind, W, Wo)
cfg = bench.est_config(hip, ds, kind, W, Wo)
clones = []
for _ in range(B):
    e = capi.Estimator(hip, cfg)
    e.copy_snapshot_of(est0)
    e.restore()
    clones.append(e)
batch = capi.EstimatorBatch(hip, clones)
batch.solve_restored(2)
import time  # noqa: E402
t0 = time.perf_counter()
reps = batch.solve_restored(steps)
dt = time.perf_counter() - t0
clk = batch.clock()
print(f"B {B}: {B * steps / dt:.0f} solves/s, {1e3 * dt / steps:.3f} ms per batch step; iterations {reps[0].iterations}, residuals {reps[0].n_lidar_residuals}")
print({k: round(v, 3) for k, v in clk.items()})
if os.environ.get("LIO_DEBUG_DIGEST"):   # phase stamps of the last marginalization (window 0 and the last one) on stderr
    batch.stage_digest(9)
batch.close()
