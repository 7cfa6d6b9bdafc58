#!/usr/bin/env python
"""The batched moments probe alone (SURVEY.md 8(d) ii) for counter runs: B copies of the bench window's lidar factors through ONE
launch of the moments kernel, in the form LIO_MOMENTS selects (mfma | valu; unset = by chunks per wave).  Usage:
batched_moments.py [B ...]  (default 1 8 64 512).  Prints one line per B."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (load order: torch first)

import bench  # noqa: E402
from lio_amd import capi  # noqa: E402

hip = capi.load_hip()
ds = bench.make_dataset("outdoor", 15)
clouds, _ = bench.feature_clouds(hip, ds)
est = bench.make_estimator(hip, ds, clouds, "outdoor", 15, 5)
bench.one_step(est)
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64, 512]:
    ms, b = est.bench_batched_moments(B, 20)
    print(f"B {B:4d}  form {os.environ.get('LIO_MOMENTS', 'auto')}  {ms * 1e3:9.2f} us per launch pair  {b / (ms * 1e-3) / 1e9:8.1f} GB/s algorithmic "
          f"({b / (ms * 1e-3) / 8e12:.3f} of 8 TB/s)  {b / 60.0 * 684.0 / (ms * 1e-3) / 1e12:6.2f} TFLOP/s MFMA-equivalent")
