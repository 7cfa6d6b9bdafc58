"""Generates tests/golden/ref_estimator_states.npz: per case of tests/ref_state_util.py the buffers of THE REFERENCE'S OWN Estimator.cc
(oracle/_ref/libref_estimator.so) after one laser message of the replay (B), the next message exactly as it was fed (M: the IMU batch, the
/compact_data message, the stamp) and what the reference had after it (C) — all from ONE run per case.  See tests/ref_state_util.py for
what it is for.  Build container only.   python tests/golden/make_ref_estimator_states.py [case ...]"""
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


class Recording(ref_est_util.RefEstimator):
    """the reference estimator, keeping what it is fed between two laser messages"""

    def __init__(self, lib, cfg):
        super().__init__(lib, cfg)
        self.batch, self.last = [], None

    def process_imu(self, dt, acc, gyr, stamp):
        self.batch.append(np.concatenate([[dt], np.asarray(acc, float), np.asarray(gyr, float), [stamp]]))
        super().process_imu(dt, acc, gyr, stamp)

    def process_compact(self, compact, stamp):
        self.last = (np.array(self.batch).reshape(-1, 8), np.array(compact, np.float32).reshape(-1, 4), float(stamp))
        self.batch = []
        return super().process_compact(compact, stamp)


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liblio_oracle.so", "ref"], check=True)
    orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    ref = ref_est_util.load()
    only = sys.argv[1:]
    out = {}
    if only:
        old = np.load(su.STATES)
        out = {k: old[k] for k in old.files if k.split("/")[0] not in only}
    for case in (only or su.STEPS):
        for k, v in dump_case(orc, ref, case).items():
            out[case + "/" + k] = v
    np.savez_compressed(su.STATES, **out)
    print({k: v.shape for k, v in out.items() if "stack" in k or k.endswith("/C/solve") or k.endswith("compact")})


def dump_case(orc, ref, CASE):
    from replay_util import run_from_zero

    c = cases.CASES[CASE]
    STEP_B = su.STEPS[CASE]
    out, state = {}, dict(s=-1)

    def configure(cfg):
        for k, v in c["cfg"].items():
            setattr(cfg, k, v)

    def on_step(rp, k, e):
        est = rp.est
        if not est.stage()["inited"]:
            return
        state["s"] += 1
        if state["s"] == STEP_B:
            pivot = c["W"] - c["Wo"]
            for key, v in est.state_dump().items():
                if key.startswith("stack") and int(key[5:]) <= pivot:
                    continue                  # after the next push these are the slots behind the pivot: spent local maps, never read again
                out["B/" + key] = np.asarray(v)
        if state["s"] == STEP_B + 1:        # the message that was just processed, and what the reference has after it (same run: two runs
            imu, compact, stamp = est.last  # of the reference part ways at the 1e-15 level and drift)
            rep, w, pr, lm = e["report"], est.get_window(), est.prior(), est.local_map()
            out.update({"M/imu": imu, "M/compact": compact, "M/stamp": np.array(stamp),
                        "C/Ps": w["Ps"], "C/Rs": w["Rs"], "C/Vs": w["Vs"], "C/Bas": w["Bas"], "C/Bgs": w["Bgs"],
                        "C/lb": np.concatenate([w["q_lb"], w["t_lb"]]).astype(float), "C/JtJ": pr["JtJ"], "C/Jtr": pr["Jtr"], "C/x0": pr["x0"],
                        "C/solve": np.array([rep.iterations, rep.termination, rep.n_lidar_residuals, rep.initial_cost, rep.final_cost], float),
                        "C/trace": np.asarray(rep.cost_trace[:11], float),
                        "C/local_map": np.concatenate([[lm.shape[0]], lm[:, :3].astype(float).sum(axis=0)])})

    n = c["n_sweeps"]
    run_from_zero(orc, n, W=c["W"], Wo=c["Wo"], init_window_factor=c["iwf"], odom_io=c["io"], kind=c["kind"], configure=configure, on_step=on_step,
                  est_factory=lambda cfg: Recording(ref, cfg), sweeps=cases.sweeps_of(c["kind"], n))
    return out


if __name__ == "__main__":
    main()
