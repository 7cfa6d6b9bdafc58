"""ROS-free replay of the reference's two nodes through the C-ABI.

    processor_node : PointProcessor::Process                              (src/processor_node.cc:83-96)
    estimator_node : PointOdometry (thread A) -> /compact_data -> Estimator::ProcessEstimation (thread B)
                                                                          (src/estimator_node.cc:142-153)

`Replay` reproduces the message pairing of MeasurementManager::GetMeasurements (MeasurementManager.cc:54-108: every IMU
message before the laser stamp plus ONE after it, which stays in the buffer) and the IMU loop of
Estimator::ProcessEstimation (Estimator.cc:2688-2730: samples up to the laser stamp are integrated as they are, the one
after it is linearly interpolated AT the laser stamp).  It holds no algorithm of its own: every step is a C-ABI call.
"""
from __future__ import annotations

import collections

import numpy as np

from .capi import Estimator, EstConfig, LioLib, PointOdometry, PointProcessor, TransformF


class Replay:
    tap = None   # (class default: subclasses that build only the pairing buffers need not set it)
    def __init__(self, lib: LioLib, cfg: EstConfig, lidar, odom_io: int = 2, msg_time_delay: float = 0.0, scan_period: float = 0.1, tap=None):
        self.lib = lib
        self.pp = PointProcessor(lib, lidar.lower_deg, lidar.upper_deg, lidar.rings)
        self.od = PointOdometry(lib, scan_period, odom_io, 25, False)
        self.est = Estimator(lib, cfg)
        self.cfg = cfg
        self.odom_io = odom_io
        self.delay = msg_time_delay
        self.imu_buf = collections.deque()      # (t, acc[3], gyr[3])
        self.compact_buf = collections.deque()  # (stamp, compact cloud)
        self.curr_time = -1.0
        self.imu_last_time = -1.0
        self.odom_frame_count = 0
        self.odom_started = False
        self.odom_enabled = True
        self.log = []                           # one dict per processed compact message
        self.tap = tap                          # tap("imu", t, acc, gyr) / tap("compact", stamp, cloud): every raw message as the ROS callbacks see it

    # ---- processor_node + PointOdometry::Process / PublishResults (PointOdometry.cc:294-683, 685-770)
    def add_sweep(self, points: np.ndarray, stamp: float):
        self.pp.process(points)
        cl = [self.pp.cloud(w) for w in (1, 2, 3, 4)]
        r = self.od.process(*cl)
        if not self.odom_started:   # the first sweep only initialises the odometry (:302-310)
            self.odom_started = True
            return
        self.odom_frame_count += 1
        if self.odom_io < 2 or self.odom_frame_count % self.odom_io == 1:
            T = TransformF.make(*r["T_sum"])
            corner, surf = self.od.last_cloud(0), self.od.last_cloud(1)
            full = np.zeros((0, 4), np.float32)  # full_cloud_ is only republished for display
            self.compact_buf.append((stamp, self.lib.compact_encode(T, corner, surf, full)))
            if self.tap:
                self.tap("compact", stamp, self.compact_buf[-1][1])
        self._drain()

    def add_imu(self, t: float, acc, gyr):
        if float(t) <= self.imu_last_time:          # "imu message in disorder!": dropped (MeasurementManager.cc:111-115)
            return
        self.imu_last_time = float(t)
        self.imu_buf.append((float(t), np.asarray(acc, float), np.asarray(gyr, float)))
        if self.tap:
            self.tap("imu", float(t), self.imu_buf[-1][1], self.imu_buf[-1][2])
        self._drain()

    # ---- MeasurementManager::GetMeasurements + Estimator::ProcessEstimation
    def _drain(self):
        while self.imu_buf and self.compact_buf:
            stamp, compact = self.compact_buf[0]
            t_laser = stamp + self.delay
            if self.imu_buf[-1][0] <= t_laser:
                return                              # wait for imu
            if self.imu_buf[0][0] >= t_laser:
                self.compact_buf.popleft()          # "throw compact_data, only should happen at the beginning"
                continue
            self.compact_buf.popleft()
            batch = []
            while self.imu_buf[0][0] < t_laser:
                batch.append(self.imu_buf.popleft())
            batch.append(self.imu_buf[0])           # one message after the laser stamp; it stays in the buffer
            self._process(batch, stamp, compact)

    def _process(self, batch, stamp, compact):
        acc = np.zeros(3)
        gyr = np.zeros(3)
        t_laser = stamp + self.delay
        for t, a, g in batch:
            if t <= t_laser:
                if self.curr_time < 0:
                    self.curr_time = t
                dt = t - self.curr_time
                self.curr_time = t
                acc, gyr = a, g
                self.est.process_imu(dt, acc, gyr, t)
            else:
                dt_1, dt_2 = t_laser - self.curr_time, t - t_laser
                self.curr_time = t_laser
                w1, w2 = dt_2 / (dt_1 + dt_2), dt_1 / (dt_1 + dt_2)
                acc, gyr = w1 * acc + w2 * a, w1 * gyr + w2 * g
                self.est.process_imu(dt_1, acc, gyr, t)
        T_to_init, rep = self.est.process_compact(compact, stamp)
        st = self.est.stage()
        if st["inited"] and self.odom_enabled and (self.cfg.enable_deskew or self.cfg.cutoff_deskew):
            self.od.enable(False)                   # the /enable_odom service call (Estimator.cc:549-558)
            self.odom_enabled = False
        self.log.append(dict(stamp=stamp, event=st["event"], inited=st["inited"], T_to_init=T_to_init, report=rep, window=None))
        return st
