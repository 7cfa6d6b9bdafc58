"""Sweeps for the infer_start_ori tests (PointProcessor.cc:348-387): one synthetic scan turned about z so that the start
azimuth drifts by a constant step per sweep, with stray leading returns at chosen sweeps, and a float64 model of the filter."""
import numpy as np


def azimuth(x, y):
    a = 2 * np.pi - np.arctan2(y, x)
    return a - 2 * np.pi if a >= 2 * np.pi else a


def norm_rad(r):
    r = np.fmod(r + np.pi, 2 * np.pi)
    if r < 0:
        r += 2 * np.pi
    return r - np.pi


def make_sweeps(scan, n_sweeps, step_rad, stray_at=(), stray_turn=2.0):
    """-> list of (n, 4) float32 scans.  Sweep k is the scan turned by -k * step about z (its start azimuth grows by `step`);
    sweeps in `stray_at` get one extra leading return `stray_turn` rad further round, which is what a dropped packet does to
    the first azimuth of a sweep."""
    scan = np.asarray(scan, np.float32)
    ok = np.isfinite(scan[:, :3]).all(1)
    scan = scan[ok]
    out = []
    for k in range(n_sweeps):
        a = -k * step_rad
        c, s = np.cos(a), np.sin(a)
        pts = scan.copy()
        pts[:, 0] = (c * scan[:, 0] - s * scan[:, 1]).astype(np.float32)
        pts[:, 1] = (s * scan[:, 0] + c * scan[:, 1]).astype(np.float32)
        if k in stray_at:
            b = -stray_turn
            cb, sb = np.cos(b), np.sin(b)
            lead = pts[:1].copy()
            lead[0, 0] = np.float32(cb * pts[0, 0] - sb * pts[0, 1])
            lead[0, 1] = np.float32(sb * pts[0, 0] + cb * pts[0, 1])
            pts = np.concatenate([lead, pts], 0)
        out.append(np.ascontiguousarray(pts))
    return out


class FilterModel:
    """The filter from its description: `used` / `seen` histories of ten; a jump of more than rad_diff from the last used
    value is replaced by last used + mean used step (wrapped to [0, 2 pi)); when the nine seen steps and the mean seen step all
    agree with the mean used step within 0.05 rad, ring 0's first azimuth is taken."""

    def __init__(self, rad_diff):
        self.used, self.seen, self.rad_diff = [], [], rad_diff

    def update(self, measured, ring0_front):
        s = measured
        self.seen = (self.seen + [s])[-10:]
        if len(self.used) >= 10:
            step_used = norm_rad(self.used[-1] - self.used[0]) / 9
            step_seen = norm_rad(self.seen[-1] - self.seen[0]) / 9
            if abs(norm_rad(s - self.used[-1])) > self.rad_diff:
                s = norm_rad(self.used[-1] + step_used)
                if s < 0:
                    s += 2 * np.pi
            even = abs(norm_rad(step_used - step_seen)) < 0.05
            for k in range(9, 0, -1):
                even = even and abs(norm_rad(self.seen[k] - self.seen[k - 1] - step_used)) < 0.05
            if even and ring0_front is not None:      # ring 0 empty: the reference dereferences an empty cloud; the value is kept
                s = ring0_front
        self.used = (self.used + [s])[-10:]
        return s
