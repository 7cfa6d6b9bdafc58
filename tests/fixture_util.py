"""SURVEY.md 8(d) config 3 inputs: the reference's own IMU fixture (test/data/imu_pose_vel_noise.txt, committed as
tests/golden/imu_noise_fixture.npz by tests/golden/make_imu_noise_fixture.py) drives the estimator; VLP-16 sweeps of
S_indoor are ray-cast from the fixture's trajectory columns."""
import os

import numpy as np

from lio_amd import capi, pipeline, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "imu_noise_fixture.npz")


def fixture_trajectory():
    rows = np.load(GOLDEN)["rows"]
    assert rows.shape == (4001, 17)
    return synth.FixtureTrajectory(rows)


def fixture_sweeps(n_sweeps=200):
    """(sweeps, pose_fn, lidar) of n motion-distorted VLP-16 sweeps along the fixture trajectory, sweep k spanning
    [0.1 k, 0.1 (k + 1)] s, plus the trajectory object."""
    traj = fixture_trajectory()
    return synth.make_sweeps("indoor", n_sweeps, t0=0.0, traj=traj), traj


def snapshot_pair(libs, traj, scan, W, Wo, keep, deskew, frame_dt=0.2, pp_lib=None):
    """Window snapshot at `scan` (scans 60 / 100 / 140): the W + 1 frames ending at t = 0.1 scan,
    states = the fixture's trajectory columns + a seeded perturbation, pre-integrations = the fixture's NOISY gyro / acc
    samples between the frames; one estimator per library on identical inputs."""
    t0 = 0.1 * scan - frame_dt * W
    assert t0 >= 0
    ds = synth.make_dataset("indoor", W + 1, frame_dt, t0=t0, traj=traj)
    pp_lib = pp_lib or libs[-1]
    clouds = [pipeline.feature_clouds(pp_lib, ds.lidar, f.scan) for f in ds.frames]
    ests = []
    for lib in libs:
        cfg = pipeline.config_indoor(lib, W, Wo)
        cfg.keep_features, cfg.cutoff_deskew, cfg.prior_factor = keep, 0 if deskew else 1, 1
        pipeline.set_extrinsic(cfg, ds)
        est = capi.Estimator(lib, cfg)
        pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=scan)
        ests.append(est)
    return ds, ests
