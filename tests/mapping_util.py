"""Shared drivers for the scan-to-map tests (CPU oracle checks and GPU parity)."""
import numpy as np

from lio_amd import pipeline, synth


def drifting_inputs(lib, kind, n_frames, frame_dt=0.1, **dataset_kw):
    """Undistorted scans along the synthetic trajectory with a drifting 'odometry' transform_sum: the scan-to-map
    step has to pull the pose back onto the map.  Returns [(corner_last, surf_last, (q_xyzw, p), p_gt)]."""
    ds = synth.make_dataset(kind, n_frames, frame_dt, **dataset_kw)
    f0 = ds.frames[0]
    R0 = f0.R_wb @ ds.R_lb.T
    p0 = f0.p_wb - R0 @ ds.t_lb
    out = []
    for k, f in enumerate(ds.frames):
        R = f.R_wb @ ds.R_lb.T
        p = f.p_wb - R @ ds.t_lb
        Rrel, prel = R0.T @ R, R0.T @ (p - p0)
        drift = np.array([0.05, -0.03, 0.02]) * k
        q = synth.quat_from_rot(Rrel @ synth.small_rot(np.array([0.002, -0.001, 0.003]) * k))
        surf, corner = pipeline.feature_clouds(lib, ds.lidar, f.scan)
        out.append((corner, surf, (q, prel + drift), prel))
    return out
