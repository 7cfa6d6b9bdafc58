"""Replay driver: turns a synthetic Dataset into calls on the C-ABI in the order estimator_node
makes them.  Works against any LioLib (the product; tests also pass the oracle)."""
from __future__ import annotations

import numpy as np

from . import synth
from .capi import Estimator, EstConfig, LioLib, PointProcessor, TransformF


def config_outdoor64(lib: LioLib, window_size=15, opt_window_size=5, parity=True) -> EstConfig:
    """config/outdoor_test_config_64.yaml with the compiled-default window (BASELINE.json: window=15)."""
    c = lib.default_est_config()
    c.window_size, c.opt_window_size = window_size, opt_window_size
    c.min_plane_dis, c.min_match_sq_dis = 0.2, 1.0
    c.corner_filter_size, c.surf_filter_size = 0.2, 0.4
    c.opt_extrinsic, c.imu_factor, c.point_distance_factor, c.prior_factor, c.marginalization_factor = 1, 1, 1, 1, 1
    c.enable_deskew, c.cutoff_deskew, c.keep_features = 1, 1, 0
    c.acc_n, c.gyr_n, c.acc_w, c.gyr_w, c.g_norm = 0.2, 0.02, 0.0002, 2.0e-5, 9.80
    c.extrinsic_stage = 1  # estimate_extrinsic: 1
    c.max_num_iterations = 10
    c.max_solver_time = -1.0 if parity else 0.10
    return c


def config_indoor(lib: LioLib, window_size=12, opt_window_size=7, parity=True) -> EstConfig:
    """config/indoor_test_config.yaml"""
    c = lib.default_est_config()
    c.window_size, c.opt_window_size = window_size, opt_window_size
    c.opt_extrinsic, c.imu_factor, c.point_distance_factor, c.prior_factor, c.marginalization_factor = 1, 1, 1, 0, 1
    c.enable_deskew, c.cutoff_deskew, c.keep_features = 1, 0, 1
    c.acc_n, c.gyr_n, c.acc_w, c.gyr_w, c.g_norm = 0.2, 0.02, 0.0002, 2.0e-5, 9.805
    c.extrinsic_stage = 1
    c.max_solver_time = -1.0 if parity else 0.10
    return c


def set_extrinsic(cfg: EstConfig, ds: synth.Dataset):
    q = synth.quat_from_rot(ds.R_lb)
    cfg.transform_lb = TransformF.make(q, ds.t_lb)


def feature_clouds(lib: LioLib, lidar: synth.Lidar, scan: np.ndarray):
    """PointProcessor::Process on one sweep -> (less_flat 'surf_last', less_sharp 'corner_last')."""
    pp = PointProcessor(lib, lidar.lower_deg, lidar.upper_deg, lidar.rings)
    pp.process(scan)
    return pp.cloud(PointProcessor.LESS_FLAT), pp.cloud(PointProcessor.LESS_SHARP)


def init_window(est: Estimator, lib: LioLib, ds: synth.Dataset, surf_clouds, pos_sigma=0.03, rot_sigma=0.005, vel_sigma=0.02, seed=3):
    """Inject frames 0..W as an already-initialised window (test hook): ground-truth
    states plus a small perturbation so the solver has work to do; stacks = VoxelGrid(surf, 0.4)."""
    W = est.W
    rng = np.random.default_rng(seed)
    n = W + 1
    Ps, Rs, Vs = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3))
    for i in range(n):
        f = ds.frames[i]
        Ps[i] = f.p_wb + rng.normal(0, pos_sigma, 3)
        Rs[i] = f.R_wb @ synth.small_rot(rng.normal(0, rot_sigma, 3))
        Vs[i] = f.v_w + rng.normal(0, vel_sigma, 3)
    Bas, Bgs = np.zeros((n, 3)), np.zeros((n, 3))
    est.set_window(Ps, Rs, Vs, Bas, Bgs, np.array([0, 0, -ds.g]))
    for i in range(n):
        est.set_surf_stack(i, lib.voxel_grid(surf_clouds[i], est.cfg.surf_filter_size))
    for i in range(1, n):
        f, fp = ds.frames[i], ds.frames[i - 1]
        acc0 = fp.imu_acc[-1] if i > 0 else ds.acc0
        gyr0 = fp.imu_gyr[-1] if i > 0 else ds.gyr0
        est.set_preintegration(i, acc0, gyr0, Bas[i], Bgs[i], f.imu_dt, f.imu_acc, f.imu_gyr)
    last = ds.frames[W]
    est.begin_frame(last.imu_acc[-1], last.imu_gyr[-1])


def feed_frame(est: Estimator, ds: synth.Dataset, k: int, surf, corner):
    """ProcessImu for every sample of frame k, then ProcessLaserOdom (solve + slide)."""
    f = ds.frames[k]
    for j in range(f.imu_dt.shape[0]):
        est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
    T = TransformF.make([0, 0, 0, 1], [0, 0, 0])
    return est.process_laser_odom(T, surf, corner, f.t)
