"""ONE estimator step from THE REFERENCE'S state (tests/ref_state_util.py): the buffers of the reference's own Estimator.cc after a laser
message of a replay — window, extrinsic, gravity, surf stacks, the raw IMU samples of every pre-integration, the prior — are injected
into the oracle's estimator through the test hooks of the C-ABI, the next message is fed exactly as the reference was fed it, and the
result is compared with what the reference itself had after that message (same run of the reference:
tests/golden/ref_estimator_states.npz).  Three states: the VLP-16 indoor configuration, the HDL-64E outdoor one, and BASELINE.json's
headline window (HDL-64E, 15 / 5, a 117 k-point local map, 67 156 plane factors).

This is the comparison the product takes on the GPU (tests/test_gpu_zz_ref_state.py); here it pins the injection itself: if any buffer
were missing or shifted by one slot the step would be off by centimetres.  Measured: positions 2e-12 .. 6e-11 m, the local map
identical, the same number of plane factors, costs to 1e-12, the new prior's JtJ 1e-11.

(Why the expected state comes from the same run as the injected ones: two runs of the reference part ways at the 1e-15 level — its
marginalization walks an address-keyed hash map — and are 1e-6 m apart a few solves later, so files from different runs cannot be mixed
below that.)"""
import numpy as np

import ref_state_util as su
from window_util import rot_angle


import pytest


@pytest.mark.parametrize("case", list(su.STEPS))
def test_one_step_from_the_reference_state(oracle, case):
    est, rep, C = su.one_step(oracle, case)
    w, pr, lm = est.get_window(), est.prior(), est.local_map()
    assert np.abs(w["Ps"] - C["Ps"]).max() <= 1e-9 and max(rot_angle(a, b) for a, b in zip(w["Rs"], C["Rs"])) <= 3e-8
    assert np.abs(w["Vs"] - C["Vs"]).max() <= 1e-8 and np.abs(w["Bas"] - C["Bas"]).max() <= 1e-8 and np.abs(w["Bgs"] - C["Bgs"]).max() <= 1e-9
    np.testing.assert_allclose(np.concatenate([w["q_lb"], w["t_lb"]]), C["lb"], atol=1e-7)
    it, term, n_lidar, c0, c1 = C["solve"]
    assert (rep.iterations, rep.termination, rep.n_lidar_residuals) == (int(it), int(term), int(n_lidar))
    np.testing.assert_allclose([rep.initial_cost, rep.final_cost], [c0, c1], rtol=1e-9)
    np.testing.assert_allclose(rep.cost_trace[:int(it) + 1], C["trace"][:int(it) + 1], rtol=1e-9)
    assert lm.shape[0] == int(C["local_map"][0])
    np.testing.assert_allclose(lm[:, :3].astype(float).sum(axis=0), C["local_map"][1:], rtol=1e-12, atol=1e-9)
    assert pr["n"] == C["JtJ"].shape[0] and np.abs(pr["JtJ"] - C["JtJ"]).max() <= 1e-8 * np.abs(C["JtJ"]).max()
    assert np.abs(pr["x0"] - C["x0"]).max() <= 1e-8
