#!/usr/bin/env python
"""Wall time of the REFERENCE's own Estimator.cc (oracle/_ref/libref_estimator.so: src/imu_processor/Estimator.cc and what it links
against, compiled where it lies under /root/reference against the stand-in headers of oracle/ref_shim) per /compact_data message of
the headline replay (tests/ref_est_cases.py "outdoor64_15_5": HDL-64E sweeps, window 15 / opt window 5), beside the oracle's on the
same messages.  Build container only (the reference tree does not travel to the GPU box).  The stand-ins matter for what the number
means: pcl::KdTreeFLANN is an exact brute-force search, Eigen's decompositions and ceres::Solve forward to the oracle's restatements
— so this is the reference's CONTROL FLOW and factor code on top of those, not its third-party libraries.
Usage: python tools/time_ref_estimator.py [case] -> one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
from lio_amd import capi  # noqa: E402
import ref_est_cases as cases  # noqa: E402
import ref_est_util  # noqa: E402


class Timed:
    """wraps an estimator: wall time of every process_compact"""

    def __init__(self, inner):
        self.inner, self.ms = inner, []

    def process_compact(self, compact, stamp):
        t = time.perf_counter()
        r = self.inner.process_compact(compact, stamp)
        self.ms.append((time.perf_counter() - t) * 1e3)
        return r

    def __getattr__(self, k):
        return getattr(self.inner, k)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "outdoor64_15_5"
    orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    ref = ref_est_util.load()
    out = {"case": name, "host": os.uname().nodename, "cores": os.cpu_count()}
    for label, factory in (("reference_Estimator_cc", lambda cfg: ref_est_util.RefEstimator(ref, cfg)), ("oracle", lambda cfg: capi.Estimator(orc, cfg))):
        holder = {}

        def make(cfg, factory=factory, holder=holder):
            holder["est"] = Timed(factory(cfg))
            return holder["est"]

        rows = cases.run_case(orc, name, est_factory=make, features_of=cases.ref_features if label.startswith("reference") else cases.oracle_features(cases.CASES[name]["W"], cases.CASES[name]["Wo"]))
        ev = [str(r["event"]) for r in rows]
        ms = np.asarray(holder["est"].ms)
        solved = np.asarray([m for m, e in zip(ms, ev) if e == "solved"])
        out[label] = {"messages": len(ev), "solved_steps": int(len(solved)), "ms_per_solved_message_median": round(float(np.median(solved)), 1) if len(solved) else None,
                      "ms_per_solved_message_min": round(float(solved.min()), 1) if len(solved) else None,
                      "n_lidar_last": int(rows[-1]["n_lidar"]) if "n_lidar" in rows[-1] else None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
