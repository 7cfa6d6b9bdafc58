"""Generates tests/golden/ref_estimator_run.npz by running the REFERENCE's own Estimator (oracle/_ref/libref_estimator.so, built by
`make -C oracle ref` from /root/reference/src/imu_processor/Estimator.cc and what it links against) over the replays of
tests/ref_est_cases.py, and the reference's MeasurementManager::GetMeasurements over the message schedules of tests/ref_mm_cases.
Build container only: /root/reference does not exist on the GPU box.   python tests/golden/make_ref_estimator_run.py [case ...]
(all cases: about 20 min, most of it the exact nearest-neighbour search that stands in for FLANN on the HDL-64E sequences)"""
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


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liblio_oracle.so", "ref"], check=True)
    orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    ref = ref_est_util.load()
    path = os.path.join(HERE, "ref_estimator_run.npz")
    only = sys.argv[1:]                      # case names: regenerate just those and keep the rest of the committed file
    out = {}
    if only:
        old = np.load(path)
        out = {k: old[k] for k in old.files if k.split("/")[0] not in only}
    for name in (only or cases.CASES):
        if name == "mm":
            continue
        rows = cases.run_case(orc, name, est_factory=lambda cfg: ref_est_util.RefEstimator(ref, cfg), features_of=cases.ref_features)
        print(name, [r["event"] for r in rows])
        for k, v in cases.pack(rows).items():
            out[name + "/" + k] = v
    if not only or "mm" in only:
        for name, (delay, msgs) in cases_mm().items():
            out["mm/" + name] = np.asarray(ref_est_util.mm_pairings(ref, delay, msgs), float).reshape(-1, 5)
    np.savez_compressed(path, **out)


def cases_mm():
    import ref_mm_cases

    return ref_mm_cases.CASES


if __name__ == "__main__":
    main()
