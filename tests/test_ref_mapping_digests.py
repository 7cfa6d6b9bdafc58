"""The oracle's scan-to-map stage against THE REFERENCE'S OWN PointMapping.cc (SURVEY.md §8 (f) 2, (a) a26, a8 decode).

tests/golden/ref_mapping_digests.json holds, per frame of the sequences in tests/ref_map_cases.py, what hyye/lio-mapping's
src/point_processor/PointMapping.cc produces when compiled where it lies against the stand-ins of oracle/ref_shim (`make -C oracle
ref`): every frame enters as a /compact_data message through the reference's own CompactDataHandler, then Process() —
TransformAssociateToMap, the shifting of the 21 x 21 x 11 window of 50 m cubes, the FOV selection, the stack / map assembly,
OptimizeTransformTobeMapped (5-NN, line and plane fits, scores, 6 x 6 system, degeneracy branch, update, termination),
TransformUpdate and UpdateMapDatabase.  Stored: transform_tobe_mapped_ / transform_aft_mapped_ bit patterns, digests of the
down-sampled stacks and the from-map clouds, the window centre, the valid-cube list and a digest of those cubes' contents.
Stood in and therefore NOT independently pinned: pcl::VoxelGrid (forwards to the oracle's restatement), the kd-tree (exact search),
Eigen's ColPivHouseholderQR / SelfAdjointEigenSolver (forwarded to the oracle's; the facade honours Eigen's lower-triangle rule,
which the reference's line fit relies on).

Equality is bit for bit.  The product's GPU path is held to the oracle by tests/test_gpu_mapping.py (1e-4 m / 1e-4 rad, cube
contents up to counted voxel-face flips)."""
import json
import os

import pytest

from ref_map_cases import cases, replay_lib

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mapping_digests.json")))
NAMES = ["indoor_sequence", "outdoor_sequence", "window_shift", "frozen_after_imu_init"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_point_mapping_equals_the_reference(oracle, name):
    frames = dict(cases(oracle))[name]
    rows = replay_lib(oracle, frames)
    want = GOLD[name]
    assert len(rows) == len(want)
    for k, (a, b) in enumerate(zip(rows, want)):
        assert a == b, (name, k, [key for key in a if a[key] != b[key]])
    assert all(len(r["valid"]) > 100 for r in rows)


def test_committed_digests_are_what_the_reference_produces(tmp_path):
    """Build container only: rebuild oracle/_ref from /root/reference and regenerate."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/src/point_processor"):
        pytest.skip("the reference tree is not on this machine")
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"], check=True)
    gen = os.path.join(root, "tests", "golden", "make_ref_mapping_digests.py")
    out = str(tmp_path / "d.json")
    code = open(gen).read().replace('path = os.path.join(HERE, "ref_mapping_digests.json")', f"path = {out!r}").replace("__file__", repr(gen))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True)
    assert json.load(open(out)) == GOLD
