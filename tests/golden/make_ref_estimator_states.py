"""Generates tests/golden/ref_estimator_states.npz: the buffers of THE REFERENCE'S OWN Estimator.cc (oracle/_ref/libref_estimator.so) after
two consecutive laser messages of the `indoor` replay — see tests/ref_state_util.py for what is stored and what it is for.  Build container
only.   python tests/golden/make_ref_estimator_states.py"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))

from lio_amd import capi  # noqa: E402
import ref_est_cases as cases  # noqa: E402
import ref_est_util  # noqa: E402
import ref_state_util as su  # noqa: E402


def main():
    from replay_util import run_from_zero

    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liblio_oracle.so", "ref"], check=True)
    orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    ref = ref_est_util.load()
    out = {}
    for case in su.STEPS:
        for k, v in dump_case(orc, ref, case).items():
            out[case + "/" + k] = v
    np.savez_compressed(su.STATES, **out)
    print({k: v.shape for k, v in out.items() if "stack" in k or k.endswith("/C/solve")})


def dump_case(orc, ref, CASE):
    from replay_util import run_from_zero

    c = cases.CASES[CASE]
    STEP_A, STEP_B = su.STEPS[CASE]
    out, state = {}, dict(s=-1)

    def configure(cfg):
        for k, v in c["cfg"].items():
            setattr(cfg, k, v)

    def on_step(rp, k, e):
        if not rp.est.stage()["inited"]:
            return
        state["s"] += 1
        for tag, step in (("A", STEP_A), ("B", STEP_B)):
            if state["s"] == step:
                pivot = c["W"] - c["Wo"]
                for key, v in rp.est.state_dump().items():
                    if key.startswith("stack") and int(key[5:]) <= pivot:
                        continue                  # after the next push these are the slots behind the pivot: spent local maps, never read again
                    out[tag + "/" + key] = np.asarray(v)

        if state["s"] == STEP_B + 1:        # what the reference itself had one message after B — from the SAME run (two runs of
            rep, w, pr = e["report"], rp.est.get_window(), rp.est.prior()   # the reference part ways at the 1e-15 level and drift)
            lm = rp.est.local_map()
            out.update({"C/Ps": w["Ps"], "C/Rs": w["Rs"], "C/Vs": w["Vs"], "C/Bas": w["Bas"], "C/Bgs": w["Bgs"],
                        "C/lb": np.concatenate([w["q_lb"], w["t_lb"]]).astype(float), "C/JtJ": pr["JtJ"], "C/Jtr": pr["Jtr"], "C/x0": pr["x0"],
                        "C/solve": np.array([rep.iterations, rep.termination, rep.n_lidar_residuals, rep.initial_cost, rep.final_cost], float),
                        "C/trace": np.asarray(rep.cost_trace[:11], float),
                        "C/local_map": np.concatenate([[lm.shape[0]], lm[:, :3].astype(float).sum(axis=0)])})

    n = c["n_sweeps"]
    run_from_zero(orc, n, W=c["W"], Wo=c["Wo"], init_window_factor=c["iwf"], odom_io=c["io"], kind=c["kind"], configure=configure, on_step=on_step,
                  est_factory=lambda cfg: ref_est_util.RefEstimator(ref, cfg), sweeps=cases.sweeps_of(c["kind"], n))
    return out


if __name__ == "__main__":
    main()
