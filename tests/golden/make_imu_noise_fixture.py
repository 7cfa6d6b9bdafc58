#!/usr/bin/env python
"""Generates tests/golden/imu_noise_fixture.npz from the reference's second hot-path fixture,
/root/reference/test/data/imu_pose_vel_noise.txt (loader: include/utils/LoadVirtual.h:84-106; columns
t qw qx qy qz px py pz vx vy vz gx gy gz ax ay az + 6 uninitialised bias columns that are dropped): 4001 rows = 20 s at
200 Hz of a simulated IMU (gyro / acc columns WITH noise, std 0.21 rad/s / 0.27 m/s^2) and its ground-truth trajectory
(pose / velocity columns, identical to imu_pose_vel.txt).  SURVEY.md 8(d) config 3 drives the full LIO estimator with this
stream; tests/test_gpu_fixture_replay.py does.  Runs only in the build container (the reference is not on the GPU box);
the file is committed (float64, exactly the parsed values)."""
import os

import numpy as np

src = "/root/reference/test/data/imu_pose_vel_noise.txt"
rows = np.loadtxt(src)[:, :17]
assert rows.shape == (4001, 17) and abs(rows[1, 0] - 0.005) < 1e-12
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "imu_noise_fixture.npz")
np.savez_compressed(out, rows=rows, source=np.array("hyye/lio-mapping test/data/imu_pose_vel_noise.txt, all 4001 rows, cols 0..16"))
print(out, rows.shape, os.path.getsize(out))
