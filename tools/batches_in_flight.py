#!/usr/bin/env python
"""K lio_est_batch objects of B / K windows each, solved from K host threads at the same time (one caller thread per handle, as the
header asks): does a second batch in flight fill the first one's host phases, sync bubbles and latency chains?
Usage: batches_in_flight.py B K [steps] [loop_groups]   (loop_groups 0 = the library's default)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
import torch  # noqa: E402,F401  (torch before the product library: both bring a HIP runtime)

import bench  # noqa: E402
from lio_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
per = B // K
batches = [capi.EstimatorBatch(hip, clones[k * per:(k + 1) * per]) for k in range(K)]
for b in batches:
    if groups > 0:
        b.set_option("loop_groups", groups)
    b.solve_restored(2)
reps = [None] * K


def run(k):
    reps[k] = batches[k].solve_restored(steps)


best = None
for trial in range(3):
    th = [threading.Thread(target=run, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
dig = [b.stage_digest(s) for b in batches for s in range(len(capi.EstimatorBatch.STAGES))]
ns = len(capi.EstimatorBatch.STAGES)
same = all(bool((dig[k * ns + s] == dig[s][0]).all()) for k in range(K) for s in range(ns))
its = {(r.iterations, r.n_lidar_residuals, r.final_cost) for rr in reps for r in rr}
print(f"B {B} as {K} x {per} (loop_groups {groups or 'default'}): {per * K * steps / best:.0f} solves/s, {1e3 * best / steps:.3f} ms per step of all batches; "
      f"all digests equal {same}; distinct reports {len(its)}")
for b in batches:
    b.close()
