#!/usr/bin/env python
"""The PRODUCT against the reference's own Estimator, step by step (measurement, not yet a test: written after round 3's GPU budget
was spent, so its bounds have not been calibrated on hardware — next round's first gpurun call).

Replays the cases of tests/ref_est_cases.py through the HIP library with the same teacher forcing tests/test_ref_estimator_run.py
applies to the oracle: after every laser message of an initialised estimator the window, the extrinsic and the marginalization prior
are overwritten with what the REFERENCE's Estimator.cc produced (tests/golden/ref_estimator_run.npz), so every solve starts from the
reference's state.  Prints, per step, the product's gap to the reference (and the oracle's, run beside it) in position, rotation,
velocity, biases, extrinsic, number of lidar factors, iteration count, final cost and the prior's JtJ.
    gpurun -- 'python tools/gpu_ref_estimator_gaps.py [case ...] > gpurun_out/ref_estimator_gaps.txt'
The front end (PointProcessor / PointOdometry -> /compact_data) is the oracle's on both sides, as in the golden file: what differs is
the estimator alone (before the initialisation that includes the product's scan-to-map stage on the GPU).
What to expect: the step at which the estimator initialises cannot be forced beforehand, and the product reaches it through its own
fp32 scan-to-map chain (millimetres away from the reference's, tests/test_gpu_end_to_end.py); its SlideWindow then moves the stacked
clouds with those states, and those clouds stay in the window for W more steps — so the first W steps after the initialisation carry
that offset in their local map even though the states are forced, and only the later ones are a clean one-step comparison."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))

from lio_amd import capi  # noqa: E402
import ref_est_cases as cases  # noqa: E402
from window_util import rot_angle  # noqa: E402


def run(front, est_lib, name, ref):
    """like cases.run_case, with only the calls the product is known to serve"""
    from replay_util import run_from_zero

    c = cases.CASES[name]
    rows = []

    def configure(cfg):
        for k, v in c["cfg"].items():
            setattr(cfg, k, v)

    def on_step(rp, k, e):
        est = rp.est
        st = est.stage()
        r = dict(event=e["event"], inited=st["inited"])
        if st["inited"]:
            rep, w = e["report"], est.get_window()
            r.update(Ps=w["Ps"], Rs=w["Rs"], Vs=w["Vs"], Bas=w["Bas"], Bgs=w["Bgs"], lb=np.concatenate([w["q_lb"], w["t_lb"]]).astype(float),
                     iterations=rep.iterations, n_lidar=rep.n_lidar_residuals, final_cost=rep.final_cost)
            pr = est.prior()
            if pr is not None:
                r.update(JtJ=pr["JtJ"])
            f = ref[len(rows)]
            if f.get("inited") is not None and bool(f["inited"]):
                est.set_window(f["Ps"], f["Rs"], f["Vs"], f["Bas"], f["Bgs"], f["g_vec"])
                est.set_extrinsic(f["lb"][:4], f["lb"][4:])
                if "prior_jac" in f and pr is not None and int(f["prior_n"]) == pr["n"]:
                    est.set_prior_factor(dict(n=int(f["prior_n"]), lin_jac=f["prior_jac"], lin_res=f["prior_res"], x0=f["x0"]))
        rows.append(r)

    run_from_zero(front, c["n_sweeps"], W=c["W"], Wo=c["Wo"], init_window_factor=c["iwf"], odom_io=c["io"], kind=c["kind"], configure=configure,
                  on_step=on_step, est_factory=(lambda cfg: capi.Estimator(est_lib, cfg)) if est_lib is not front else None)
    return rows


def gaps(a, b):
    out = dict(dP=float(np.abs(a["Ps"] - b["Ps"]).max()), dR=max(rot_angle(x, y) for x, y in zip(a["Rs"], b["Rs"])),
               dV=float(np.abs(a["Vs"] - b["Vs"]).max()), dBa=float(np.abs(a["Bas"] - b["Bas"]).max()), dBg=float(np.abs(a["Bgs"] - b["Bgs"]).max()),
               dlb=float(np.abs(a["lb"] - b["lb"]).max()), dn=int(a["n_lidar"]) - int(b["n_lidar"]), it=(int(a["iterations"]), int(b["iterations"])),
               dcost=abs(float(a["final_cost"]) - float(b["final_cost"])) / float(b["final_cost"]))
    if "JtJ" in a and "JtJ" in b and a["JtJ"].shape == b["JtJ"].shape:
        out["dJtJ"] = float(np.abs(a["JtJ"] - b["JtJ"]).max() / np.abs(b["JtJ"]).max())
    return out


def main():
    oracle = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    hip = capi.load_hip()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_estimator_run.npz"))
    for name in (sys.argv[1:] or ["indoor", "outdoor64", "indoor_12_7"]):
        ref = cases.unpack(gold, name)
        got = {"hip": run(oracle, hip, name, ref), "oracle": run(oracle, oracle, name, ref)}
        for side, rows in got.items():
            print(f"== {name}: {side} vs the reference's Estimator.cc; events equal: {[r['event'] for r in rows] == [r['event'] for r in ref]}")
            s = -1
            for a, b in zip(rows, ref):
                if not (a["inited"] and bool(b["inited"])):
                    continue
                s += 1
                print(f"  step {s}: " + "  ".join(f"{k} {v:.2e}" if isinstance(v, float) else f"{k} {v}" for k, v in gaps(a, b).items()))


if __name__ == "__main__":
    main()
