"""CPU check of the SURVEY.md 8(d) config 3 inputs: the committed copy of the reference's noisy IMU fixture is intact, and
the ORACLE driven by it (90 of the 200 VLP-16 scans: scan-to-scan, scan-to-map, IMU initialisation, the first solves)
initialises and tracks the fixture's ground-truth columns.  The GPU suite replays all 200 scans on both back ends."""
import numpy as np

from fixture_util import fixture_sweeps, fixture_trajectory
from replay_util import run_from_zero, window_vs_truth


def test_noise_fixture_is_the_reference_file():
    tr = fixture_trajectory()
    r = tr.rows
    assert r.shape == (4001, 17) and abs(tr.h - 0.005) < 1e-12
    # first row of test/data/imu_pose_vel_noise.txt as printed there (t qw qx qy qz px py pz vx vy vz gx gy gz ax ay az)
    np.testing.assert_allclose(r[0], [0, 0.99875, 0.0499792, 0, 0, 20, 5, 5, -0, 6.28319, 3.14159, 0.267575, 0.0584136, -0.172399, -1.3718,
                                      1.50144, 9.731], rtol=0, atol=0)
    np.testing.assert_allclose(np.linalg.norm(r[:, 1:5], axis=1), 1.0, atol=2e-6)
    # gyro / acc carry the simulated sensor noise: differentiating the pose columns gives a far smoother rate
    assert 0.15 < np.std(np.diff(r[:, 11])) / np.sqrt(2) < 0.3


def test_oracle_initialises_on_the_noise_fixture(oracle):
    W, Wo, n = 12, 7, 90
    sweeps, traj = fixture_sweeps(n)
    rp, _ = run_from_zero(oracle, n, W=W, Wo=Wo, init_window_factor=3, odom_io=2, sweeps=sweeps, traj=traj, t0=0.0)
    ev = [e["event"] for e in rp.log]
    assert ev.count("filling") == W and ev.index("initialised") == 3 * (W + 1) - 1
    assert ev[-1] == "solved" and ev.count("solved") >= 4
    errs, _ = window_vs_truth(rp, traj, W)
    # a handful of solves after initialisation the window still holds the pre-init frames (0.6 s apart, states from the
    # initialiser on 0.27 m/s^2 accelerometer noise): decimetres here, 2 cm once the window has turned over (the 200-scan GPU test)
    assert errs[:, 0].max() < 0.4 and errs[:, 1].max() < 2.0, errs
    rep = rp.log[-1]["report"]
    assert rep.n_lidar_residuals > 30000 and rep.laser_odom_kz == 0
