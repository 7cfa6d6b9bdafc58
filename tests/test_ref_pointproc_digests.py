"""The oracle's PointProcessor against THE REFERENCE'S OWN PointProcessor.cc.

tests/golden/ref_pointproc_digests.json holds digests of what hyye/lio-mapping's src/point_processor/PointProcessor.cc — compiled
where it lies against container-only stand-ins for PCL / ROS (oracle/ref_shim, `make -C oracle ref`) — produces on the seeded
sweeps of tests/ref_pp_cases.py: the ring-ordered cloud with ring + relative time, the sharp / less-sharp / flat picks and the
voxel-filtered less-flat cloud, for VLP-16 and HDL-64E sweeps, the ring-field overload, a non-default configuration and a
24-sweep run of the start-azimuth filter.  Equality of digests = every coordinate and every intensity equal bit for bit, in the
same order.  (pcl::VoxelGrid in the stand-in forwards to the oracle's own restatement: the less-flat digest pins the SET handed
to the filter and the rel-time recompute behind it, not the filter.)

The GPU twin of this file is tests/test_gpu_parity.py, which holds the product to the oracle bit for bit on the same kind of
sweeps; tests/test_gpu_ref_pointproc.py replays these very cases through the product."""
import json
import os

import numpy as np
import pytest

from lio_amd import capi
from ref_pp_cases import CLOUDS, cases, digest

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pointproc_digests.json")))
ORDER = [capi.PointProcessor.RINGS, capi.PointProcessor.SHARP, capi.PointProcessor.LESS_SHARP, capi.PointProcessor.FLAT, capi.PointProcessor.LESS_FLAT]


def replay(lib, name, lid, cfg_over, sweeps):
    cfg = capi.PPConfig()
    lib.dll.lio_pp_default_config(cfg)
    for k, v in cfg_over.items():
        if k != "uneven":
            setattr(cfg, k, v)
    pp = capi.PointProcessor(lib, lid.lower_deg, lid.upper_deg, lid.rings, cfg)
    rows = []
    for scan, ring in sweeps:
        pp.process(scan, ring)
        row = {c: digest(pp.cloud(w)) for c, w in zip(CLOUDS, ORDER)}
        row["intensity_scans"] = digest(pp.ring_intensity())   # the public intensity_scans' intensity channel (coordinates = laser_scans')
        rows.append(row)
    return rows


CASES = cases()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_point_processor_equals_the_reference(oracle, case):
    name, lid, cfg_over, sweeps = case
    rows = replay(oracle, name, lid, cfg_over, sweeps)
    assert len(rows) == len(GOLD[name])
    for k, (got, want) in enumerate(zip(rows, GOLD[name])):
        assert got == want, (name, k, {c: (got[c], want[c]) for c in list(CLOUDS) + ["intensity_scans"] if got[c] != want[c]})
    assert all(int(r["sharp"].split(":")[0]) > 0 and int(r["less_flat"].split(":")[0]) > 1000 for r in rows)


def test_committed_digests_are_what_the_reference_produces(tmp_path):
    """Build container only: rebuild oracle/_ref from /root/reference and regenerate the digests."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/src/point_processor"):
        pytest.skip("the reference tree is not on this machine")
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"], check=True)
    gen = os.path.join(root, "tests", "golden", "make_ref_pointproc_digests.py")
    out = str(tmp_path / "d.json")
    code = open(gen).read().replace('path = os.path.join(HERE, "ref_pointproc_digests.json")', f"path = {out!r}").replace("__file__", repr(gen))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True)
    assert json.load(open(out)) == GOLD
