"""The degeneracy branch of the scan-to-scan odometry (PointOdometry.cc:584-615) against THE REFERENCE'S OWN
PointOdometry.cc on the degenerate scenes of tests/degenerate_util.py (golden: tests/golden/ref_degenerate_odometry.json, made by
make_ref_degenerate_odometry.py from oracle/_ref/libref_odometry.so).

What the reference does there and what can be compared: in the first iteration the eigenvalues of the 6 x 6 normal matrix below 10 are
counted (kz), and from then on every update is multiplied by mat_P = V2 V^-1, V2 being the eigenvector matrix with its first kz ROWS
zeroed — which is diag(0 .. 0, 1 .. 1) whatever V is: the first kz COMPONENTS of the update (rotation x, y, z, then translation) are
dropped.  The oracle (and the product) apply exactly that; the reference's fp32 product carries rounding noise of the order of 1e-7.
On the regular scene (two poles, kz = 0) the two agree bit for bit.  On the singular ones (kz = 1 and 3) the dropped rotation
components receive no update on either side — that is the reading of A.6 being pinned — while the components that stay come out of a singular
6 x 6 solve and are rounding noise in any implementation, the reference included; they are not compared."""
import json
import os

import numpy as np
import pytest

import degenerate_util as D
from lio_amd import capi
from ref_odom_cases import bits

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_degenerate_odometry.json")))


def _f(b):
    return np.array(b, np.uint32).view(np.float32).astype(float)


@pytest.mark.parametrize("name", list(D.ODOMETRY_SCENES))
def test_degenerate_scan_to_scan_against_the_reference(oracle, name):
    cls, singular = D.odometry_sweeps(oracle, name, 3)
    od = capi.PointOdometry(oracle, 0.1, 2, 25, False)
    seen = []
    for k, cl in enumerate(cls):
        r = od.process(*cl)
        mine = np.concatenate([r["T_es"][0], r["T_es"][1]])
        want = _f(GOLD[name][k])
        if k == 0:
            assert bits(mine) == GOLD[name][k]
            continue
        kz = int(r["kz"])
        seen.append(kz)
        if kz == 0:
            assert bits(mine) == GOLD[name][k], (name, k)
        else:
            assert singular and 1 <= kz <= 3
            # the dropped rotation components never receive an update (q = x y z w).  With all three dropped the rotation stays the
            # identity; with one dropped, the composition of the 25 updates about the two kept axes leaves only their second-order
            # commutators in it (measured 4e-6 against kept components of 2e-4 .. 8e-2), on both sides
            for q in (mine, want):
                if kz == 3:
                    assert np.all(np.abs(q[:3]) < 1e-6) and abs(q[3] - 1.0) < 1e-6, (name, k, kz, q)
                else:
                    assert np.all(np.abs(q[:kz]) <= 1e-5) and np.abs(q[kz:3]).max() >= 1e-4, (name, k, kz, q)
            # the kept rotation components did move on both sides (the mask is per component, not all-or-nothing)
            if kz < 3:
                assert np.abs(mine[kz:3]).max() > 1e-5 and np.abs(want[kz:3]).max() > 1e-5
    assert (max(seen) > 0) == singular
