"""Scenes and drivers for the degeneracy-branch tests (Estimator.cc:1308-1339, PointOdometry.cc:584-615,
PointMapping.cc:650-680, MapBuilder.cc:930-960).

The branch: in round 0 of every 6x6 Gauss-Newton loop the eigenvalues of AtA are compared with a threshold (10 for
scan-to-scan, 100 elsewhere); kz = the number of leading (smallest) eigenvalues below it, and from then on the first kz
COMPONENTS of every update are zeroed (the reference's mat_P = V2 V^-1 is diag(0..0, 1..1), A.6).

Scene families (lio_amd/synth.py), each with a member on either side of a threshold:
  * ground plane only (+ n poles): x, y, yaw unobservable; 0 / 1 / 2 poles straddle the scan-to-scan threshold;
  * corridor (+ a low wall across it): x unobservable; the wall's height moves the smallest eigenvalue from ~40-75
    (kz = 1) to ~125 (kz = 0) around the estimator's / scan-to-map's threshold of 100.

What can be compared.  With kz > 0 the UNMASKED components of the update still come from the (near-)singular 6x6 solve:
`mat_AtA.colPivHouseholderQr().solve(mat_AtB)` divides rounding noise by a tiny pivot, so in the truly singular scenes
(ground plane) those components are noise in ANY implementation, the reference included — tests there compare kz, the
masked components (exactly zero on both sides) and nothing that depends on the noise.  In the ill-conditioned but regular
scenes (corridor: smallest eigenvalue 40-130 against a largest of 2e5) everything is compared as usual."""
import numpy as np

from lio_amd import capi, pipeline, synth

ESTIMATOR_SCENES = {
    # name: (scene factory, range noise, expected kz of the oracle, singular?)
    "ground": (lambda: synth.scene_ground_only(), 0.02, 2, True),
    "corridor_below_threshold": (lambda: synth.scene_corridor(cap_height=1.0), 0.02, 1, False),
    "corridor_above_threshold": (lambda: synth.scene_corridor(cap_height=3.0), 0.02, 0, False),
}

ODOMETRY_SCENES = {
    "ground": (lambda: synth.scene_ground_only(0), 0.003, True),
    "ground_one_pole": (lambda: synth.scene_ground_only(1), 0.003, True),
    "ground_two_poles": (lambda: synth.scene_ground_only(2), 0.003, False),
}

MAPPING_SCENES = {
    "ground": (lambda: synth.scene_ground_only(), 0.003, True),
    "corridor_below_threshold": (lambda: synth.scene_corridor(cap_height=0.0), 0.003, False),
    "corridor_above_threshold": (lambda: synth.scene_corridor(cap_height=3.0), 0.003, False),
}


def estimator_pair(libs, scene_name, W=6, Wo=3, pp_lib=None):
    """One estimator per library on the same injected window of the named scene (indoor VLP-16 configuration)."""
    factory, sigma, kz, singular = ESTIMATOR_SCENES[scene_name]
    ds = synth.make_dataset("indoor", W + 1, 0.2, scene=factory(), traj=synth.traj_corridor(), range_sigma=sigma)
    pp_lib = pp_lib or libs[-1]
    clouds = [pipeline.feature_clouds(pp_lib, ds.lidar, f.scan) for f in ds.frames]
    ests = []
    for lib in libs:
        cfg = pipeline.config_indoor(lib, W, Wo)
        cfg.keep_features, cfg.prior_factor, cfg.cutoff_deskew = 0, 1, 1
        pipeline.set_extrinsic(cfg, ds)
        est = capi.Estimator(lib, cfg)
        pipeline.init_window(est, lib, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01)
        ests.append(est)
    return ds, ests, kz, singular


def odometry_sweeps(pp_lib, scene_name, n_sweeps=3):
    """[(sharp, less_sharp, flat, less_flat)] of consecutive motion-distorted sweeps of the named scene."""
    factory, sigma, singular = ODOMETRY_SCENES[scene_name]
    sweeps, _, lid = synth.make_sweeps("indoor", n_sweeps, scene=factory(), traj=synth.traj_corridor(), range_sigma=sigma)
    out = []
    for sw in sweeps:
        pp = capi.PointProcessor(pp_lib, lid.lower_deg, lid.upper_deg, lid.rings)
        pp.process(sw)
        out.append([pp.cloud(w) for w in (1, 2, 3, 4)])
    return out, singular


def relative_rotation_vec(q_prev, q_next):
    """vector part of q_prev^-1 * q_next (x,y,z,w quaternions): DeltaQ's theta/2 when q_next = q_prev * DeltaQ(theta)."""
    x0, y0, z0, w0 = (float(v) for v in q_prev)
    x1, y1, z1, w1 = (float(v) for v in q_next)
    n = x0 * x0 + y0 * y0 + z0 * z0 + w0 * w0
    x0, y0, z0 = -x0 / n, -y0 / n, -z0 / n
    w0 = w0 / n
    return np.array([w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1, w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1])


def masked_rotation_components(trace, q_start, kz):
    """max |component| of the per-iteration rotation increments over the first min(kz, 3) axes, and over the rest."""
    q = np.asarray(q_start, np.float64)
    masked, free = 0.0, 0.0
    for row in trace:
        v = relative_rotation_vec(q, row[:4])
        k = min(kz, 3)
        if k:
            masked = max(masked, float(np.max(np.abs(v[:k]))))
        if k < 3:
            free = max(free, float(np.max(np.abs(v[k:]))))
        q = np.asarray(row[:4], np.float64)
    return masked, free
