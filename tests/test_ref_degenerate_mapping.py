"""The degeneracy branch of the scan-to-map stage (PointMapping.cc:650-680) against THE REFERENCE'S OWN PointMapping.cc
on the degenerate scenes of tests/degenerate_util.py (golden: tests/golden/ref_degenerate_mapping.json, made by
make_ref_degenerate_mapping.py from oracle/_ref/libref_mapping.so): a corridor whose cross wall puts the smallest eigenvalue of the
6 x 6 normal matrix above (kz = 0) or below (kz = 1) the threshold of 100, and a bare ground plane (kz = 3, then 2).

With kz = 0 the oracle equals the reference bit for bit.  With kz > 0 the reference multiplies every update by mat_P = V2 V^-1 computed
in fp32 — diag(0 .. 0, 1 .. 1) in exact arithmetic, which is what the oracle and the product apply (the first kz COMPONENTS of the
update are dropped) — so the two agree to the rounding noise of that product: measured 2e-7, bound 2e-6, with equal iteration counts
implied (a different count would move the pose by far more)."""
import json
import os

import numpy as np
import pytest

import degenerate_util as D
from lio_amd import capi
from ref_odom_cases import bits

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_degenerate_mapping.json")))


def _frames(oracle, name):
    from lio_amd import synth
    from mapping_util import drifting_inputs

    factory, sigma, _ = D.MAPPING_SCENES[name]
    return drifting_inputs(oracle, "indoor", len(GOLD[name]), scene=factory(), traj=synth.traj_corridor(), range_sigma=sigma)


def _f(b):
    return np.array(b, np.uint32).view(np.float32).astype(float)


@pytest.mark.parametrize("name", list(D.MAPPING_SCENES))
def test_degenerate_scan_to_map_against_the_reference(oracle, name):
    m = capi.PointMapping(oracle)
    seen, worst = [], 0.0
    for k, (corner, surf, T_sum, _) in enumerate(_frames(oracle, name)):
        r = m.process(corner, surf, T_sum)
        q, p = m.transform_tobe_mapped()
        mine = np.concatenate([q, p])
        seen.append(int(r["kz"]))
        if r["kz"] == 0:
            assert bits(mine) == GOLD[name][k], (name, k)
        else:
            gap = float(np.abs(mine.astype(float) - _f(GOLD[name][k])).max())
            worst = max(worst, gap)
            assert gap <= 2e-6, (name, k, gap)
    want_kz = {"ground": [0, 3, 2], "corridor_below_threshold": [0, 1, 1], "corridor_above_threshold": [0, 0, 0]}[name]
    assert seen == want_kz, seen
    print(name, "kz", seen, "worst gap with kz > 0", worst)
