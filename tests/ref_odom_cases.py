"""The sweep sequences on which the reference's own PointOdometry was run for tests/golden/ref_odometry_digests.json — shared by
the generator (tests/golden/make_ref_odometry_digests.py, build container only) and tests/test_ref_odometry_digests.py.
Inputs are regenerated (seeded synthetic sweeps through the oracle's PointProcessor, itself pinned to the reference's); only
the transforms (as bit patterns) and digests of the clouds are committed."""
import numpy as np

from lio_amd import capi, synth
from ref_pp_cases import digest


def bits(a):
    return [int(v) for v in np.ascontiguousarray(a, np.float32).view(np.uint32)]


def cases():
    """-> list of (name, kind, n_sweeps, io_ratio, no_deskew, disable_after)"""
    return [("indoor_io2", "indoor", 4, 2, 0, None), ("outdoor_io3", "outdoor", 3, 3, 0, None), ("indoor_no_deskew", "indoor", 3, 2, 1, None),
            ("indoor_packer_after_1", "indoor", 3, 2, 0, 1), ("indoor_io1_long", "indoor", 8, 1, 0, None), ("outdoor_no_deskew", "outdoor", 3, 2, 1, None)]


def full_cloud_for(case, cl, k):
    """What the reference is handed as /full_cloud.  While the odometry runs it publishes TransformToEnd(full cloud) evaluated AFTER
    transform_es_.rot.normalize() (PointOdometry.cc:660, 725) — no entry point of the C-ABI computes that, so those cases carry an
    empty full cloud; in packer mode the full cloud passes through untouched and the less-sharp cloud stands in for it."""
    disable_after = case[5]
    if disable_after is not None and k >= disable_after:
        return cl[1]
    return np.zeros((0, 4), np.float32)


def feature_clouds(oracle, lid, sweep):
    pp = capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings)
    pp.process(sweep)
    return [pp.cloud(w) for w in (1, 2, 3, 4)]


def replay_oracle(lib, oracle, case):
    """the same sequence through `lib`'s lio_odom_* entry points -> the rows the generator stores for the reference"""
    name, kind, n, io, no_deskew, disable_after = case
    sweeps, _, lid = synth.make_sweeps(kind, n)
    od = capi.PointOdometry(lib, 0.1, io, 25, bool(no_deskew))
    rows, frame = [], 0
    for k, sw in enumerate(sweeps):
        cl = feature_clouds(oracle, lid, sw)
        if disable_after is not None and k == disable_after:
            od.enable(False)
        r = od.process(*cl)
        corner, surf = od.last_cloud(0), od.last_cloud(1)
        row = {"T_es": bits(np.concatenate([r["T_es"][0], r["T_es"][1]])), "T_sum": bits(np.concatenate([r["T_sum"][0], r["T_sum"][1]])),
               "last_corner": digest(corner), "last_surf": digest(surf), "compact": "none"}
        if k > 0:
            frame += 1
            if io < 2 or frame % io == 1:
                T = capi.TransformF.make(*r["T_sum"])
                row["compact"] = digest(lib.compact_encode(T, corner, surf, full_cloud_for(case, cl, k)))
        rows.append(row)
    return rows
