"""The batched keyframe refinement (BASELINE.json configs[4], lio_kf_batch_*) against THE REFERENCE'S OWN Gauss-Newton loops run one keyframe
at a time: PointMapping::OptimizeTransformTobeMapped (6-DoF) and MapBuilder::OptimizeMap (4-DoF) from the sources where they lie
(oracle/ref_mapbuilder.cc: ref_kf_refine; golden: tests/golden/ref_kf_refine.json, made by make_ref_kf_refine.py), on the local maps,
down-sampled stacks and perturbed initial poses of tests/ref_kf_cases.py.

6-DoF: the pose the oracle's batch returns for every keyframe equals the reference's bit for bit.  4-DoF: within 2e-6 — every such
optimisation is degenerate by design and the reference's fp32 V2 V^-1 projection carries rounding noise (tests/test_ref_mapbuilder_digests.py).
The product's batch is held to the oracle's, keyframe by keyframe, by tests/test_gpu_kf_batch.py."""
import json
import os

import numpy as np
import pytest

import ref_kf_cases as kc
from kf_util import load
from lio_amd import capi
from ref_odom_cases import bits

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kf_refine.json")))


@pytest.mark.parametrize("name", list(kc.CASES))
def test_keyframe_refinement_against_the_reference_loops(oracle, name):
    four_dof = kc.CASES[name][3]
    maps, kfs = kc.inputs(oracle, name)
    r = load(capi.KeyframeBatch(oracle, map_builder=four_dof, enable_4d=four_dof), maps, kfs).refine()
    assert len(kfs) == len(GOLD[name])
    worst = 0.0
    for k in range(len(kfs)):
        mine = np.concatenate([r["q"][k], r["p"][k]]).astype(np.float32)
        if not four_dof:
            assert int(r["kz"][k]) == 0 and bits(mine) == GOLD[name][k], (name, k)
        else:
            want = np.array(GOLD[name][k], np.uint32).view(np.float32)
            gap = float(np.abs(mine.astype(float) - want.astype(float)).max())
            worst = max(worst, gap)
            assert int(r["kz"][k]) == 2 and gap <= 2e-6, (name, k, gap)
    assert int(np.min(r["iterations"])) >= 1
    print(name, "keyframes", len(kfs), "worst gap", worst)
