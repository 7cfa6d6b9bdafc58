"""Message schedules for MeasurementManager::GetMeasurements (MeasurementManager.cc:54-108): name -> (msg_time_delay, messages), a
message being ("imu", stamp) or ("laser", stamp) in arrival order.  GetMeasurements is polled after every message, as the condition
variable of Estimator::ProcessEstimation does."""
import numpy as np


def _regular(n_laser, imu_rate, laser_dt, t0, imu_first, laser_first, jitter_seed=None):
    rng = np.random.default_rng(jitter_seed) if jitter_seed is not None else None
    msgs = []
    for k in range(n_laser):
        msgs.append(("laser", laser_first + k * laser_dt + (rng.uniform(-0.004, 0.004) if rng else 0.0)))
    n_imu = int((laser_first + n_laser * laser_dt - imu_first) * imu_rate) + 3
    for j in range(n_imu):
        msgs.append(("imu", imu_first + j / imu_rate + (rng.uniform(-0.0004, 0.0004) if rng else 0.0)))
    # arrival order: by stamp, laser messages 30 ms late (they are processed sweeps)
    msgs.sort(key=lambda m: m[1] + (0.03 if m[0] == "laser" else 0.0))
    return msgs


CASES = {
    "regular": (0.0, _regular(8, 200.0, 0.1, 1.0, 1.0, 1.1)),
    # laser messages that precede every IMU message are thrown away
    "laser_first": (0.0, _regular(8, 200.0, 0.1, 1.0, 1.25, 1.1)),
    # an IMU stamp that coincides with a laser stamp stays in the buffer (strict <) and is the "one after"
    "coincident": (0.0, _regular(6, 100.0, 0.1, 2.0, 2.0, 2.1)),
    "delayed": (0.013, _regular(8, 200.0, 0.1, 1.0, 1.0, 1.1, jitter_seed=5)),
    # laser messages arrive in a burst after a long IMU stretch: several pairings from one poll
    "burst": (0.0, [("imu", 1.0 + 0.005 * j) for j in range(120)] + [("laser", 1.1), ("laser", 1.2), ("laser", 1.3), ("laser", 1.7)]
              + [("imu", 1.6 + 0.005 * j) for j in range(40)]),
    # an IMU message out of order is dropped by ImuHandler (MeasurementManager.cc:111-115)
    "disorder": (0.0, [("imu", 1.00), ("imu", 1.01), ("imu", 1.005), ("imu", 1.02), ("laser", 1.015), ("imu", 1.03), ("imu", 1.04)]),
}
