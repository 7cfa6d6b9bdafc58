"""The PRODUCT taking ONE estimator step from THE REFERENCE'S state, on the GPU (tests/ref_state_util.py; the oracle takes the same step in
tests/test_ref_estimator_state.py and lands 1e-11 m from the reference).

The buffers of the reference's own Estimator.cc after a laser message of a replay (window, extrinsic, gravity, surf stacks, raw IMU
samples of every pre-integration, prior; three states, BASELINE.json's headline window among them) are injected through the test hooks of the C-ABI — the calls the injected-window parity tests
use (lio_amd.pipeline.init_window / feed_frame) —, the next message is fed (ProcessImu per sample, ProcessLaserOdom -> BuildLocalMap ->
CalculateFeatures / CalculateLaserOdom -> SolveOptimization -> marginalization -> SlideWindow), and what comes out is compared with what the
reference's code produced from the same state.  Bounds: those of the product-vs-oracle contract tests (tests/window_util.py: 1e-4 m /
1e-4 rad, the north star), the same iteration count, the number of plane factors within 1 %, costs within 1e-3.

Written after round 3's GPU budget was spent: dry-run on the CPU with the oracle in the product's place only.  (Named to sort last.)"""
import numpy as np
import pytest

import ref_state_util as su
from window_util import assert_windows_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", list(su.STEPS))
def test_product_one_step_from_the_reference_state(hip, case):
    est, rep, C = su.one_step(hip, case)
    w = est.get_window()
    it, term, n_lidar, c0, c1 = C["solve"]
    print(case, "product vs the reference's Estimator.cc, one step from its state: |dP|", float(np.abs(w["Ps"] - C["Ps"]).max()), "|dV|",
          float(np.abs(w["Vs"] - C["Vs"]).max()), "iterations", rep.iterations, int(it), "plane factors", rep.n_lidar_residuals, int(n_lidar),
          "final cost", rep.final_cost, float(c1))
    assert_windows_close(w, dict(Ps=C["Ps"], Rs=C["Rs"], Vs=C["Vs"], Bas=C["Bas"], Bgs=C["Bgs"]))     # 1e-4 m / 1e-4 rad
    assert rep.iterations == int(it)
    assert abs(rep.n_lidar_residuals - int(n_lidar)) <= 0.01 * n_lidar
    assert abs(rep.final_cost - c1) <= 1e-3 * c1
    pr = est.prior()
    assert pr is not None and pr["n"] == C["JtJ"].shape[0]
    flips = abs(rep.n_lidar_residuals - int(n_lidar))
    assert np.abs(pr["JtJ"] - C["JtJ"]).max() <= (1e-4 + 20.0 * flips / n_lidar) * np.abs(C["JtJ"]).max()
