#!/usr/bin/env python
"""Generates tests/golden/imu_fixture_slice.npz from the reference's only hot-path fixture,
/root/reference/test/data/imu_pose_vel.txt (loader: include/utils/LoadVirtual.h:84-106; columns
t qw qx qy qz px py pz vx vy vz gx gy gz ax ay az + 6 uninitialised bias columns that are dropped).
Runs only in the build container (the reference is not on the GPU box); the slice is committed.
Rows 0..400 = the first 2 s at 200 Hz."""
import os

import numpy as np

src = "/root/reference/test/data/imu_pose_vel.txt"
rows = np.loadtxt(src, max_rows=401)[:, :17]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "imu_fixture_slice.npz")
np.savez_compressed(out, rows=rows, source=np.array("hyye/lio-mapping test/data/imu_pose_vel.txt rows 0..400, cols 0..16"))
print(out, rows.shape)
