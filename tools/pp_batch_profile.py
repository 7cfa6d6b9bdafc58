#!/usr/bin/env python
"""B HDL-64E sweeps resident in HBM through lio_pp_process_batch_device (one launch chain over all sweeps): the command the
PointProcessor batch profiles are taken of.  Usage: pp_batch_profile.py B [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lio_amd import capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hip = capi.load_hip()
ds = synth.make_dataset("outdoor", 4, 0.1)
lid = ds.lidar
scans = [f.scan for f in ds.frames]
dev = [torch.from_numpy(np.ascontiguousarray(s, np.float32)).cuda() for s in scans]
torch.cuda.synchronize()
hs = [capi.PointProcessor(hip, lid.lower_deg, lid.upper_deg, lid.rings) for _ in range(B)]
ptr = [dev[i % 4].data_ptr() for i in range(B)]
n = [dev[i % 4].shape[0] for i in range(B)]
capi.PointProcessor.process_batch_device(hs, ptr, n)
t = time.perf_counter()
for _ in range(reps):
    capi.PointProcessor.process_batch_device(hs, ptr, n)
dt = time.perf_counter() - t
npts = float(np.mean(n))
rate = B * reps / dt
print(f"B {B}: {rate:.0f} sweeps/s, {1e3 * dt / reps:.3f} ms per call, {rate * npts * 40 / 1e9:.1f} GB/s of 40 B per point = {rate * npts * 40 / 8e12:.4f} of 8 TB/s; "
      f"picks of sweep 0: {[int(hip.dll.lio_pp_count(hs[0].h, w)) for w in range(5)]}")
