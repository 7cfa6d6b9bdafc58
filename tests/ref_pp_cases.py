"""The sweeps on which the reference's own PointProcessor was run for tests/golden/ref_pointproc_digests.json — shared by the
generator (tests/golden/make_ref_pointproc_digests.py, build container only) and the tests that replay them through the oracle
and the product.  Inputs are regenerated (seeded synthetic scans), only digests of the outputs are committed."""
import hashlib

import numpy as np

from lio_amd import synth
from pp_util import ring_field
from start_ori_util import make_sweeps

CLOUDS = ["laser_scans", "sharp", "less_sharp", "flat", "less_flat"]      # index k of the reference wrapper = {5, 1, 2, 3, 4}[k]


def digest(a):
    a = np.ascontiguousarray(a, np.float32)
    return f"{a.shape[0]}:" + hashlib.sha256(a.tobytes()).hexdigest()[:32]


def cases():
    """-> list of (name, lidar, config overrides, [(scan, ring or None), ...])"""
    out = []
    for kind in ("indoor", "outdoor"):
        ds = synth.make_dataset(kind, 2, 0.1)
        out.append((f"{kind}_elevation", ds.lidar, {}, [(f.scan, None) for f in ds.frames]))
    ds = synth.make_dataset("indoor", 1, 0.1)
    scan = ds.frames[0].scan
    out.append(("indoor_ring_field", ds.lidar, {"uneven": 1}, [(scan, ring_field(scan, ds.lidar))]))
    # the other two sensor presets of processor_node.cc:66-74: sensor_type 32 = PointProcessor(-30.67, 10.67, 32) (elevation formula),
    # sensor_type 320 = PointProcessor(-25, 15, 32, uneven = true) (ring field + swept range)
    lid32 = synth.Lidar(32, -30.67, 10.67, 1800)
    ds32 = synth.make_dataset("indoor", 1, 0.1, lidar=lid32)
    out.append(("preset32_elevation", lid32, {}, [(ds32.frames[0].scan, None)]))
    lid320 = synth.Lidar(32, -25.0, 15.0, 1800)
    ds320 = synth.make_dataset("indoor", 1, 0.1, lidar=lid320)
    out.append(("preset320_ring_field", lid320, {"uneven": 1}, [(ds320.frames[0].scan, ring_field(ds320.frames[0].scan, lid320))]))
    # other thresholds / quotas / subregion counts than the defaults
    out.append(("indoor_other_config", ds.lidar, {"num_scan_subregions": 6, "max_corner_sharp": 3, "max_corner_less_sharp": 12, "max_surf_flat": 5,
                                                   "surf_curv_th": 0.25, "less_flat_filter_size": 0.3}, [(scan, None)]))
    # the start-azimuth filter: 24 sweeps with a drifting start azimuth and stray leading returns at three of them
    sweeps = make_sweeps(scan, 24, 0.03, stray_at=(14, 15, 20))
    out.append(("indoor_infer_start_ori", ds.lidar, {"infer_start_ori": 1}, [(s, None) for s in sweeps]))
    return out
