"""Helpers for the PointProcessor tests: ring ids of a synthetic scan as a sensor with a ring field would report them."""
import numpy as np


def ring_field(scan, lidar):
    """Ring of each return from its elevation (exact: the synthetic range noise acts along the ray). NaN returns get 0."""
    x, y, z = (scan[:, k].astype(np.float64) for k in range(3))
    el = np.degrees(np.arctan2(z, np.hypot(x, y)))
    r = np.rint((el - lidar.lower_deg) / (lidar.upper_deg - lidar.lower_deg) * (lidar.rings - 1))
    r = np.where(np.isfinite(r), r, 0)
    return np.clip(r, 0, lidar.rings - 1).astype(np.uint16)
