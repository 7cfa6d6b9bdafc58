"""The PRODUCT, from t = 0 on the GPU, against what the REFERENCE's own Estimator.cc produced on the same sweeps
(tests/golden/ref_estimator_run.npz, case `indoor` of tests/ref_est_cases.py: the reference's sources compiled where they lie,
see oracle/ref_estimator.cc and tests/test_ref_estimator_run.py, which holds the oracle to that file at 1e-9 .. 3e-5 m).

Free running and through the product's own front end (PointProcessor / PointOdometry on the GPU), so the bounds are those of
tests/test_gpu_end_to_end.py, for the reason given there: the scan-to-map loop's termination test makes the chained pipeline
discontinuous at the millimetre level, and a 1e-7 m difference in the odometry moves the initialisation by millimetres.  What this
adds to that test is the right-hand side: the numbers compared against were computed by the reference's code, not by the oracle.
(The step-by-step, teacher-forced comparison of the product with the same file is tools/gpu_ref_estimator_gaps.py — to become a test
once its bounds have been measured on hardware.)"""
import os

import numpy as np
import pytest

import ref_est_cases as cases
from lio_amd import synth
from replay_util import run_from_zero

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_estimator_run.npz")


def test_replay_from_zero_matches_the_reference_estimator(hip):
    c = cases.CASES["indoor"]
    gold = cases.unpack(np.load(GOLDEN), "indoor")
    n = 26                                       # 13 laser messages: the stretch tests/test_gpu_end_to_end.py covers against the oracle
    rp, _ = run_from_zero(hip, n, W=c["W"], Wo=c["Wo"], init_window_factor=c["iwf"], odom_io=c["io"], sweeps=synth.make_sweeps("indoor", n))
    ev = [e["event"] for e in rp.log]
    assert len(ev) >= 12 and ev == [r["event"] for r in gold[:len(ev)]] and "initialised" in ev
    worst_T, worst = 0.0, {}
    for e, r in zip(rp.log, gold):
        q, p = np.asarray(e["T_to_init"][0], float), np.asarray(e["T_to_init"][1], float)
        worst_T = max(worst_T, float(np.max(np.abs(p - r["T"][4:]))), float(min(np.max(np.abs(q - r["T"][:4])), np.max(np.abs(q + r["T"][:4])))))
        if e["window"] is None:
            continue
        for key in ("Ps", "Rs", "Vs", "Bgs"):
            worst[key] = max(worst.get(key, 0.0), float(np.max(np.abs(e["window"][key] - r[key]))))
    print("product vs the reference's Estimator.cc: worst T_to_init diff", worst_T, "worst window diffs", worst)
    assert worst_T < 0.03, worst_T
    assert worst["Ps"] < 0.15 and worst["Rs"] < 0.01 and worst["Vs"] < 0.15 and worst["Bgs"] < 3e-3, worst
    st, last = rp.est.stage(), gold[len(ev) - 1]
    np.testing.assert_allclose(st["R_WI"], last["R_WI"], atol=5e-3)
    np.testing.assert_allclose(st["g_vec"], last["g_vec"], atol=1e-6)
