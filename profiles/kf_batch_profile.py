#!/usr/bin/env python
"""Times the batched keyframe refinement alone (bench.py's `keyframe_batch` extra) — for A/B runs and rocprofv3.
Usage: python profiles/kf_batch_profile.py [n_keyframes]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
import bench  # noqa: E402
from lio_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hip = capi.LioLib(capi.HIP_LIB_PATH)
ds = bench.make_dataset("outdoor", 15)
clouds, _ = bench.feature_clouds(hip, ds)
captured = []
bench.mapping_ms_per_scan(hip, ds, clouds, capture=captured)
print(json.dumps(bench.keyframe_batch_stats(hip, captured, n)))
