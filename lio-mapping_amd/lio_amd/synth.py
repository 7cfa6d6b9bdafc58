"""Deterministic synthetic inputs for the LIO hot path.

No dataset or reference fixture is read at run time: scenes are analytic (axis-aligned boxes,
vertical cylinders, a ground plane), scans are ray-cast with numpy, and the IMU stream is the exact
derivative of an analytic trajectory of the same family as the reference's fixture
(test/data/imu_pose_vel.txt: ellipse in xy, sinusoid in z, small roll/pitch oscillation).

Frames: world z-up, gravity (0,0,-g).  `T_wb` body(IMU) in world; lidar pose follows SURVEY.md A.17:
R_wl = R_wb R_lb^T, p_wl = p_wb - R_wl t_lb.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

# ------------------------------------------------------------------------------------------------
# rotations
# ------------------------------------------------------------------------------------------------


def rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return rz @ ry @ rx


def quat_from_rot(R):
    """x,y,z,w (Shepperd)."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0, 0.0, 0.0]
        q[i] = 0.25 * s
        w = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        x, y, z = q
    return np.array([x, y, z, w])


def rot_from_quat(q):
    x, y, z, w = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def small_rot(v):
    """exp map of a rotation vector."""
    th = np.linalg.norm(v)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + math.sin(th) / th * K + (1 - math.cos(th)) / (th * th) * (K @ K)


# ------------------------------------------------------------------------------------------------
# trajectory + IMU
# ------------------------------------------------------------------------------------------------


@dataclasses.dataclass
class Trajectory:
    """p(t) = c + (rx cos(K t), ry sin(K t), rz sin(Kz t)); euler zyx = (K t + pi/2, 0.2 sin t, 0.1 cos t)."""

    rx: float = 15.0
    ry: float = 20.0
    rz: float = 1.0
    cx: float = 5.0
    cy: float = 5.0
    cz: float = 5.0
    K: float = 2 * math.pi / 20.0
    Kz: float = 10 * 2 * math.pi / 20.0
    g: float = 9.805
    ang_scale: float = 1.0

    def pos(self, t):
        return np.array([self.cx + self.rx * math.cos(self.K * t), self.cy + self.ry * math.sin(self.K * t), self.cz + self.rz * math.sin(self.Kz * t)])

    def vel(self, t):
        return np.array([-self.rx * self.K * math.sin(self.K * t), self.ry * self.K * math.cos(self.K * t), self.rz * self.Kz * math.cos(self.Kz * t)])

    def acc(self, t):
        return np.array(
            [-self.rx * self.K**2 * math.cos(self.K * t), -self.ry * self.K**2 * math.sin(self.K * t), -self.rz * self.Kz**2 * math.sin(self.Kz * t)]
        )

    def euler(self, t):
        a = self.ang_scale
        return self.K * t + math.pi / 2, a * 0.2 * math.sin(t), a * 0.1 * math.cos(t)

    def euler_rate(self, t):
        a = self.ang_scale
        return self.K, a * 0.2 * math.cos(t), -a * 0.1 * math.sin(t)

    def rot(self, t):
        return rot_zyx(*self.euler(t))

    def gyro(self, t):
        _, p, r = self.euler(t)
        yd, pd, rd = self.euler_rate(t)
        return np.array(
            [rd - yd * math.sin(p), pd * math.cos(r) + yd * math.sin(r) * math.cos(p), -pd * math.sin(r) + yd * math.cos(r) * math.cos(p)]
        )

    def accel(self, t):
        """specific force in the body frame: R^T (a - g_w), g_w = (0,0,-g)"""
        return self.rot(t).T @ (self.acc(t) + np.array([0, 0, self.g]))


class FixtureTrajectory:
    """A sampled trajectory + IMU stream in the layout of the reference's test/data/imu_pose_vel*.txt (rows at a fixed rate;
    columns t qw qx qy qz px py pz vx vy vz gx gy gz ax ay az — include/utils/LoadVirtual.h:84-106).  Poses between samples
    are interpolated (position / velocity linearly, rotation by normalised linear interpolation of the quaternions: at
    200 Hz the chord error is < 1e-7); accel / gyro return the SAMPLE at the nearest row, i.e. the recorded stream."""

    def __init__(self, rows, g=9.805):
        self.rows = np.asarray(rows, dtype=np.float64)
        self.t0 = float(self.rows[0, 0])
        self.h = float(self.rows[1, 0] - self.rows[0, 0])
        self.g = g

    def _bracket(self, t):
        u = (t - self.t0) / self.h
        i = int(np.clip(math.floor(u), 0, self.rows.shape[0] - 2))
        return i, float(np.clip(u - i, 0.0, 1.0))

    def pos(self, t):
        i, a = self._bracket(t)
        return (1 - a) * self.rows[i, 5:8] + a * self.rows[i + 1, 5:8]

    def vel(self, t):
        i, a = self._bracket(t)
        return (1 - a) * self.rows[i, 8:11] + a * self.rows[i + 1, 8:11]

    def rot(self, t):
        i, a = self._bracket(t)
        q0, q1 = self.rows[i, 1:5], self.rows[i + 1, 1:5]
        if np.dot(q0, q1) < 0:
            q1 = -q1
        q = (1 - a) * q0 + a * q1
        q = q / np.linalg.norm(q)
        return rot_from_quat(np.array([q[1], q[2], q[3], q[0]]))

    def _row(self, t):
        return int(np.clip(round((t - self.t0) / self.h), 0, self.rows.shape[0] - 1))

    def gyro(self, t):
        return self.rows[self._row(t), 11:14].copy()

    def accel(self, t):
        return self.rows[self._row(t), 14:17].copy()


# ------------------------------------------------------------------------------------------------
# scenes + ray casting
# ------------------------------------------------------------------------------------------------


@dataclasses.dataclass
class Scene:
    room: np.ndarray | None  # (2,3) min/max of an enclosing box seen from inside, or None
    boxes: np.ndarray  # (M,2,3)
    cyls: np.ndarray  # (C,5): cx, cy, r, z0, z1
    ground_z: float | None
    max_range: float


def scene_indoor(seed=20190406):
    rng = np.random.default_rng(seed)
    room = np.array([[5 - 20.0, 5 - 30.0, 0.0], [5 + 20.0, 5 + 30.0, 10.0]])
    boxes = []
    for _ in range(12):
        c = np.array([rng.uniform(-12, 22), rng.uniform(-22, 32), 0.0])
        s = np.array([rng.uniform(0.8, 3.0), rng.uniform(0.8, 3.0), rng.uniform(1.0, 6.0)])
        boxes.append([[c[0] - s[0] / 2, c[1] - s[1] / 2, 0.0], [c[0] + s[0] / 2, c[1] + s[1] / 2, s[2]]])
    cyls = []
    for _ in range(8):
        cyls.append([rng.uniform(-12, 22), rng.uniform(-22, 32), rng.uniform(0.15, 0.5), 0.0, 10.0])
    return Scene(room, np.array(boxes), np.array(cyls), None, 100.0)


def scene_outdoor(seed=64):
    rng = np.random.default_rng(seed)
    boxes = []
    for _ in range(60):
        c = np.array([rng.uniform(-85, 115), rng.uniform(-85, 115)])
        s = np.array([rng.uniform(6, 25), rng.uniform(6, 25), rng.uniform(4, 25)])
        boxes.append([[c[0] - s[0] / 2, c[1] - s[1] / 2, 0.0], [c[0] + s[0] / 2, c[1] + s[1] / 2, s[2]]])
    cyls = []
    for _ in range(80):
        cyls.append([rng.uniform(-85, 115), rng.uniform(-85, 115), rng.uniform(0.1, 0.4), 0.0, rng.uniform(3, 12)])
    return Scene(None, np.array(boxes), np.array(cyls), 0.0, 120.0)


def scene_ground_only(n_poles=0, pole_radius=0.3):
    """A single ground plane: x, y and yaw are unobservable (the leading eigenvalues of every 6x6 AtA are ~0) — the
    degeneracy branch of SURVEY.md A.6 with kz up to 3.  `n_poles` vertical cylinders 6-14 m from the origin add a few
    edge rows each: with the scan-to-scan threshold of 10 (PointOdometry.cc:584-615) zero / one / two poles give
    kz = 3 (2) / 1 / 0 on the synthetic VLP-16 sweeps, i.e. the scene family straddles that threshold."""
    ang = np.arange(n_poles) * (2 * np.pi / max(n_poles, 1)) + 0.3
    dist = 6.0 + 2.0 * np.arange(n_poles) % 5
    cyls = np.array([[d * np.cos(a), d * np.sin(a), pole_radius, 0.0, 6.0] for a, d in zip(ang, dist)]).reshape(-1, 5)
    return Scene(None, np.zeros((0, 2, 3)), cyls, 0.0, 60.0)


def scene_corridor(cap_height=0.0, cap_x=14.0, half_width=2.0, height=3.0):
    """A 4 m x 3 m corridor along x whose end walls lie beyond the sensor's range: translation along x is unobservable
    (kz = 1).  cap_height > 0 adds a low wall across the corridor at x = cap_x: the few returns on it are the only
    constraint along x, so its height places the smallest eigenvalue of AtA on either side of the degeneracy thresholds
    (100 for the estimator / scan-to-map, 10 for scan-to-scan)."""
    room = np.array([[-400.0, -half_width, 0.0], [400.0, half_width, height]])
    boxes = np.zeros((0, 2, 3))
    if cap_height > 0:
        boxes = np.array([[[cap_x, -half_width, 0.0], [cap_x + 0.5, half_width, cap_height]]])
    return Scene(room, boxes, np.zeros((0, 5)), None, 45.0)


def traj_corridor():
    """Slow motion about the corridor's axis: +-3 m along x, +-0.3 m across, +-5 cm in height, one yaw turn per 20 s."""
    return Trajectory(rx=3.0, ry=0.3, rz=0.05, cx=0.0, cy=0.0, cz=1.5, Kz=2 * math.pi / 5.0, g=9.805, ang_scale=0.2)


def raycast(scene: Scene, origin: np.ndarray, dirs: np.ndarray, chunk: int = 8192) -> np.ndarray:
    """Nearest hit distance along unit `dirs` (N,3) from `origin` ((3,) or per-ray (N,3)); inf when nothing is hit.
    Rays are independent: they are traced `chunk` at a time so that the ~10 temporaries per primitive stay in cache (1.5x faster
    on a 133 k-ray HDL-64E sweep, the same bits)."""
    n = dirs.shape[0]
    origin = np.asarray(origin, dtype=np.float64)
    if origin.ndim == 1:
        origin = np.broadcast_to(origin, (n, 3))
    if n > chunk:
        out = np.empty(n)
        for s in range(0, n, chunk):
            out[s:s + chunk] = raycast(scene, origin[s:s + chunk], dirs[s:s + chunk], chunk)
        return out
    best = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dirs
        if scene.room is not None:
            t1 = (scene.room[0] - origin) * inv
            t2 = (scene.room[1] - origin) * inv
            hi = np.maximum(t1, t2)
            tfar = np.minimum(np.minimum(hi[:, 0], hi[:, 1]), hi[:, 2])   # (column-wise: numpy's reduction over a 3-long axis is slow)
            best = np.where(tfar > 0, tfar, best)
        if scene.ground_z is not None:
            t = (scene.ground_z - origin[:, 2]) * inv[:, 2]
            ok = (t > 0) & np.isfinite(t)
            best = np.where(ok & (t < best), t, best)
        for b in scene.boxes:
            t1 = (b[0] - origin) * inv
            t2 = (b[1] - origin) * inv
            lo, hi = np.minimum(t1, t2), np.maximum(t1, t2)
            tn = np.maximum(np.maximum(lo[:, 0], lo[:, 1]), lo[:, 2])
            tf = np.minimum(np.minimum(hi[:, 0], hi[:, 1]), hi[:, 2])
            ok = (tn <= tf) & (tn > 0)
            best = np.where(ok & (tn < best), tn, best)
        for c in scene.cyls:
            ox, oy = origin[:, 0] - c[0], origin[:, 1] - c[1]
            a = dirs[:, 0] ** 2 + dirs[:, 1] ** 2
            bq = 2 * (ox * dirs[:, 0] + oy * dirs[:, 1])
            cq = ox * ox + oy * oy - c[2] ** 2
            disc = bq * bq - 4 * a * cq
            sq = np.sqrt(np.maximum(disc, 0))
            t = (-bq - sq) / (2 * a)
            z = origin[:, 2] + t * dirs[:, 2]
            ok = (disc > 0) & (t > 0) & (z >= c[3]) & (z <= c[4]) & np.isfinite(t)
            best = np.where(ok & (t < best), t, best)
    return best


@dataclasses.dataclass
class Lidar:
    rings: int
    lower_deg: float
    upper_deg: float
    n_azimuth: int

    @staticmethod
    def vlp16():
        return Lidar(16, -15.0, 15.0, 1800)

    @staticmethod
    def hdl64():
        return Lidar(64, -24.9, 2.0, 2083)  # src/processor_node.cc:71


def make_scan(scene: Scene, lidar: Lidar, R_wl: np.ndarray, p_wl: np.ndarray, seed: int, range_sigma=0.02, nan_frac=0.005, pose_fn=None,
              t_start=0.0, scan_period=0.1) -> np.ndarray:
    """One sweep in the lidar frame, azimuth-major firing order (all rings per azimuth step, clockwise),
    float32 (N,4) = x,y,z,intensity.  N = rings * n_azimuth (NaN returns kept, out-of-range dropped).
    With `pose_fn(t) -> (R_wl, p_wl)` the sweep is motion-distorted: azimuth step a is fired at
    t_start + scan_period * (a + 0.5) / n_azimuth from the pose at that instant (what a spinning lidar records)."""
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(lidar.lower_deg, lidar.upper_deg, lidar.rings))
    az = -(np.arange(lidar.n_azimuth) + 0.5) * (2 * np.pi / lidar.n_azimuth)  # clockwise: atan2 decreasing
    azg, elg = np.meshgrid(az, el, indexing="ij")  # (n_az, rings): azimuth-major
    d_l = np.stack([np.cos(elg) * np.cos(azg), np.cos(elg) * np.sin(azg), np.sin(elg)], axis=-1).reshape(-1, 3)
    if pose_fn is None:
        d_w = d_l @ R_wl.T
        rngs = raycast(scene, p_wl, d_w)
    else:
        d_w = np.empty_like(d_l)
        o_w = np.empty_like(d_l)
        for a in range(lidar.n_azimuth):
            Ra, pa = pose_fn(t_start + scan_period * (a + 0.5) / lidar.n_azimuth)
            sl = slice(a * lidar.rings, (a + 1) * lidar.rings)
            d_w[sl] = d_l[sl] @ Ra.T
            o_w[sl] = pa
        rngs = raycast(scene, o_w, d_w)
    rngs = rngs + rng.normal(0.0, range_sigma, size=rngs.shape)
    keep = np.isfinite(rngs) & (rngs > 0.5) & (rngs < scene.max_range)
    pts = d_l * rngs[:, None]
    nan_mask = rng.random(rngs.shape[0]) < nan_frac
    pts[nan_mask] = np.nan
    pts = pts[keep | nan_mask]
    out = np.zeros((pts.shape[0], 4), dtype=np.float32)
    out[:, :3] = pts.astype(np.float32)
    out[:, 3] = 10.0
    return out


# ------------------------------------------------------------------------------------------------
# window datasets
# ------------------------------------------------------------------------------------------------


@dataclasses.dataclass
class FrameData:
    t: float
    R_wb: np.ndarray
    p_wb: np.ndarray
    v_w: np.ndarray
    scan: np.ndarray  # raw sweep in the lidar frame
    imu_dt: np.ndarray  # samples in (t_prev, t]
    imu_acc: np.ndarray
    imu_gyr: np.ndarray
    imu_t: np.ndarray


@dataclasses.dataclass
class Dataset:
    frames: list
    R_lb: np.ndarray
    t_lb: np.ndarray
    g: float
    acc0: np.ndarray  # imu sample at frames[0].t
    gyr0: np.ndarray
    lidar: Lidar


def make_dataset(kind: str, n_frames: int, frame_dt: float, t0: float = 1.0, imu_rate: float = 200.0, seed: int = 7, imu_noise: bool = False,
                 lidar: Lidar | None = None, scene: Scene | None = None, traj: Trajectory | None = None, range_sigma: float = 0.02) -> Dataset:
    """kind = 'indoor' (VLP-16, S_indoor, fixture-like trajectory) or 'outdoor' (HDL-64E, S_outdoor, xy x3); `scene` / `traj`
    replace the kind's scene / trajectory (its extrinsic, gravity and lidar stay)."""
    scene_in, traj_in = scene, traj
    if kind == "indoor":
        scene, lid, traj = scene_indoor(), lidar or Lidar.vlp16(), Trajectory()
        R_lb, t_lb, g = np.eye(3), np.array([0.0, 0.0, -0.081939]), 9.805  # indoor_test_config.yaml:23-36
    elif kind == "outdoor":
        scene, lid = scene_outdoor(), lidar or Lidar.hdl64()
        traj = Trajectory(rx=45.0, ry=60.0, rz=0.3, cx=15.0, cy=15.0, cz=2.2, Kz=2 * math.pi / 5.0, g=9.80, ang_scale=0.3)
        R_lb = np.array(  # outdoor_test_config_64.yaml extrinsic (laser^R_imu)
            [[9.999976e-01, 7.553071e-04, -2.035826e-03], [-7.854027e-04, 9.998898e-01, -1.482298e-02], [2.024406e-03, 1.482454e-02, 9.998881e-01]]
        )
        # re-orthonormalise (the YAML matrix is rounded)
        u, _, vt = np.linalg.svd(R_lb)
        R_lb = u @ vt
        t_lb, g = np.array([-8.086759e-01, 3.195559e-01, -7.997231e-01]), 9.80
    else:
        raise ValueError(kind)
    scene = scene_in or scene
    traj = traj_in or traj
    rng = np.random.default_rng(seed)
    h = 1.0 / imu_rate
    frames = []
    steps = int(round(frame_dt * imu_rate))
    for k in range(n_frames):
        tk = t0 + k * frame_dt
        R_wb, p_wb = traj.rot(tk), traj.pos(tk)
        R_wl = R_wb @ R_lb.T
        p_wl = p_wb - R_wl @ t_lb
        scan = make_scan(scene, lid, R_wl, p_wl, seed=1000 + k, range_sigma=range_sigma)
        ts = tk - frame_dt + h * (np.arange(steps) + 1)
        acc = np.array([traj.accel(t) for t in ts])
        gyr = np.array([traj.gyro(t) for t in ts])
        if imu_noise:
            acc = acc + rng.normal(0, 0.02, acc.shape)
            gyr = gyr + rng.normal(0, 0.002, gyr.shape)
        frames.append(FrameData(tk, R_wb, p_wb, traj.vel(tk), scan, np.full(steps, h), acc, gyr, ts))
    return Dataset(frames, R_lb, t_lb, g, traj.accel(t0), traj.gyro(t0), lid)


def make_sweeps(kind: str, n_sweeps: int, scan_period: float = 0.1, t0: float = 1.0, lidar: Lidar | None = None, scene: Scene | None = None,
                traj: Trajectory | None = None, range_sigma: float = 0.02):
    """Consecutive motion-distorted sweeps for the scan-to-scan odometry: sweep k spans
    [t0 + k T, t0 + (k+1) T].  Returns (sweeps, lidar pose function, lidar).  `scene` / `traj` as in make_dataset."""
    scene_in, traj_in = scene, traj
    if kind == "indoor":
        scene, lid, traj = scene_indoor(), lidar or Lidar.vlp16(), Trajectory()
        R_lb, t_lb = np.eye(3), np.array([0.0, 0.0, -0.081939])
    else:
        scene, lid = scene_outdoor(), lidar or Lidar.hdl64()
        traj = Trajectory(rx=45.0, ry=60.0, rz=0.3, cx=15.0, cy=15.0, cz=2.2, Kz=2 * math.pi / 5.0, g=9.80, ang_scale=0.3)
        R_lb, t_lb = np.eye(3), np.array([-8.086759e-01, 3.195559e-01, -7.997231e-01])
    scene = scene_in or scene
    traj = traj_in or traj

    def pose_fn(t):
        R_wb = traj.rot(t)
        R_wl = R_wb @ R_lb.T
        return R_wl, traj.pos(t) - R_wl @ t_lb

    sweeps = [make_scan(scene, lid, None, None, seed=2000 + k, range_sigma=range_sigma, pose_fn=pose_fn, t_start=t0 + k * scan_period, scan_period=scan_period)
              for k in range(n_sweeps)]
    return sweeps, pose_fn, lid
