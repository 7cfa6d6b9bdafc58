"""GPU parity of the whole pipeline from t = 0 (replay through the C-ABI): same stage events, same scan-to-map poses,
same initialisation, and the same window after several sliding-window solves as the CPU oracle.

Tolerances: every stage is compared tightly on IDENTICAL inputs elsewhere (test_gpu_parity.py, test_gpu_mapping.py).
Chained over ~25 sweeps the reference algorithm itself is discontinuous at the millimetre level: the scan-to-map loop
stops when a step falls below 0.05 cm / 0.05 deg (PointMapping.cc:75-76,714), so an input difference of 1e-7 m can
change the number of rounds (measured with the ORACLE alone: feeding it the GPU odometry, which differs by 2e-7 m,
moves its scan-to-map pose by 1.3 mm and its round count from 8 to 10).  Those millimetres enter the IMU
initialisation and the window.  The chained comparison is therefore at the centimetre level, plus the requirement
that both back ends track the analytic ground truth equally well."""
import numpy as np
import pytest

from lio_amd import synth
from replay_util import run_from_zero, window_vs_truth

pytestmark = pytest.mark.gpu


def test_replay_from_zero_matches_oracle(hip, oracle):
    W, n = 6, 26
    sweeps = synth.make_sweeps("indoor", n)
    rph, traj = run_from_zero(hip, n, W=W, Wo=3, sweeps=sweeps)
    rpo, _ = run_from_zero(oracle, n, W=W, Wo=3, sweeps=sweeps)
    ev_h, ev_o = [e["event"] for e in rph.log], [e["event"] for e in rpo.log]
    assert ev_h == ev_o and "initialised" in ev_o
    worst_T = 0.0
    for eh, eo in zip(rph.log, rpo.log):
        worst_T = max(worst_T, float(np.max(np.abs(eh["T_to_init"][1] - eo["T_to_init"][1]))))
        q_h, q_o = eh["T_to_init"][0], eo["T_to_init"][0]
        worst_T = max(worst_T, float(min(np.max(np.abs(q_h - q_o)), np.max(np.abs(q_h + q_o)))))
    assert worst_T < 0.03, worst_T
    k0 = ev_o.index("initialised")
    worst = {}
    for k in range(k0, len(ev_o)):
        wh, wo = rph.log[k]["window"], rpo.log[k]["window"]
        for key in ("Ps", "Rs", "Vs", "Bgs"):
            worst[key] = max(worst.get(key, 0.0), float(np.max(np.abs(wh[key] - wo[key]))))
    print("worst T_to_init diff", worst_T, "worst window diffs", worst)
    assert worst["Ps"] < 0.15 and worst["Rs"] < 0.01 and worst["Vs"] < 0.15 and worst["Bgs"] < 3e-3, worst
    sth, sto = rph.est.stage(), rpo.est.stage()
    np.testing.assert_allclose(sth["R_WI"], sto["R_WI"], atol=5e-3)
    np.testing.assert_allclose(sth["g_vec"], sto["g_vec"], atol=1e-6)
    errs_h, _ = window_vs_truth(rph, traj, W)
    errs_o, _ = window_vs_truth(rpo, traj, W)
    assert errs_h[:, 0].max() < 0.08 and errs_h[:, 1].max() < 1.0, errs_h
    assert abs(errs_h[:, 0].max() - errs_o[:, 0].max()) < 0.03
