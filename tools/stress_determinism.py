#!/usr/bin/env python
"""Stress check of the completion-word path (csrc/dev.h: HostSignal): the bench's step (restore + SolveOptimization on the
HDL-64E window) is repeated N times; every reduction in the solve is fixed-order, so every repetition must reproduce the
first one BIT FOR BIT (cost trace, iteration count, window).  A host read that overtook the device's stores would show up as
a differing repetition.  Also runs the 4-threads / 4-windows configuration.  Usage: stress_determinism.py [N]"""
import hashlib
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (load order: torch first)

import bench
from lio_amd import capi


def fingerprint(est, rep):
    w = est.get_window()
    h = hashlib.sha1()
    for key in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
        h.update(np.ascontiguousarray(w[key]).tobytes())
    h.update(np.asarray(rep.cost_trace[: rep.iterations + 1], dtype=np.float64).tobytes())
    return (rep.iterations, rep.termination, rep.laser_odom_iterations, rep.n_lidar_residuals, h.hexdigest())


def run(est, n, out, tag):
    first, bad = None, 0
    for k in range(n):
        rep = bench.one_step(est)
        fp = fingerprint(est, rep)
        if first is None:
            first = fp
        elif fp != first:
            bad += 1
            if bad <= 3:
                print(f"[{tag}] repetition {k} differs: {fp} vs {first}", flush=True)
    out[tag] = (first, bad)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    hip = capi.load_hip()
    ds = bench.make_dataset("outdoor", 15)
    clouds, _ = bench.feature_clouds(hip, ds)
    out = {}
    est = bench.make_estimator(hip, ds, clouds, "outdoor", 15, 5)
    run(est, n, out, "single")
    ests = [bench.make_estimator(hip, ds, clouds, "outdoor", 15, 5) for _ in range(4)]
    th = [threading.Thread(target=run, args=(e, n // 4, out, f"thread{i}")) for i, e in enumerate(ests)]
    [t.start() for t in th]
    [t.join() for t in th]
    ok = all(v[1] == 0 for v in out.values()) and len({v[0] for v in out.values()}) == 1
    for k, v in sorted(out.items()):
        print(k, "differing repetitions:", v[1], "fingerprint:", v[0][:4], v[0][4][:12])
    print("STRESS", "OK" if ok else "FAILED", f"({n} + 4 x {n // 4} solves)")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
