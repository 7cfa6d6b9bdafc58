"""The replays on which the reference's own Estimator (oracle/ref_estimator.cc -> oracle/_ref/libref_estimator.so) was run for
tests/golden/ref_estimator_run.npz — shared by the generator (tests/golden/make_ref_estimator_run.py, build container only) and
tests/test_ref_estimator_run.py, which replays the ORACLE's estimator on the same messages and compares.

Every case starts at t = 0: synthetic motion-distorted sweeps + analytic IMU -> PointProcessor -> PointOdometry -> /compact_data (the
oracle's front end on both sides; it is pinned separately, tests/test_ref_{pointproc,odometry}_digests.py) -> the estimator under test:
scan-to-map while the window fills, the IMU initialisation, then sliding-window solves, free running (nothing is teacher-forced)."""
import numpy as np

from replay_util import run_from_zero

CASES = {
    # config/indoor_test_config.yaml on a VLP-16: keep_features, de-skew by the IMU, extrinsic optimised, no prior factor
    "indoor": dict(kind="indoor", n_sweeps=30, W=6, Wo=3, iwf=1, io=2, cfg={}),
    # config/outdoor_test_config_64.yaml on an HDL-64E: prior factor, cut-off de-skew, no kept features
    "outdoor64": dict(kind="outdoor", n_sweeps=24, W=6, Wo=3, iwf=1, io=2, cfg={}),
    # every second laser message only while the window fills (init_window_factor = 2), each sweep a message (io_ratio 1)
    "indoor_iwf2": dict(kind="indoor", n_sweeps=22, W=5, Wo=2, iwf=2, io=1, cfg=dict(keep_features=0, cutoff_deskew=1)),
    # the two window sizes of BASELINE.json: indoor_test_config.yaml's 12 / 7 on the VLP-16 and the compiled default 15 / 5 on the
    # HDL-64E with every third sweep a message (the headline configuration: 64 rings, ~130 k points a sweep, window 15)
    "indoor_12_7": dict(kind="indoor", n_sweeps=34, W=12, Wo=7, iwf=1, io=2, cfg={}),
    "outdoor64_15_5": dict(kind="outdoor", n_sweeps=57, W=15, Wo=5, iwf=1, io=3, cfg={}),
    # SURVEY.md 8(d) config 3 as written: the reference's own noisy IMU fixture (test/data/imu_pose_vel_noise.txt) drives the estimator,
    # VLP-16 sweeps ray-cast along its trajectory columns, indoor_test_config.yaml's 12 / 7 window, every third message while filling
    "fixture_12_7": dict(kind="indoor", n_sweeps=90, W=12, Wo=7, iwf=3, io=2, cfg={}, fixture=True),
    # the same with estimate_extrinsic = 2: on this motion the hand-eye rotation (ImuInitializer::EstimateExtrinsicRotation) converges,
    # replaces the configured lidar-IMU rotation, and the initialisation follows in the same message
    "fixture_extrinsic2": dict(kind="indoor", n_sweeps=84, W=12, Wo=7, iwf=3, io=2, cfg=dict(extrinsic_stage=2), fixture=True),
    # switches of EstimatorConfig one at a time (indoor configuration otherwise)
    "indoor_fixed_extrinsic": dict(kind="indoor", n_sweeps=20, W=6, Wo=3, iwf=1, io=2, cfg=dict(opt_extrinsic=0)),
    "indoor_no_marginalization": dict(kind="indoor", n_sweeps=20, W=6, Wo=3, iwf=1, io=2, cfg=dict(marginalization_factor=0)),
    "indoor_prior_factor": dict(kind="indoor", n_sweeps=20, W=6, Wo=3, iwf=1, io=2, cfg=dict(prior_factor=1, keep_features=0)),
    # point_distance_factor off: no lidar factor reaches the solver (IMU factors and the prior only).  The reference's marginalization
    # then keeps only the blocks its remaining factors touch (pose 1, speed-bias 1: n = 15) where the oracle and the product keep
    # their fixed layout with empty rows for the rest (n = 33) — the same information; the prior is not compared in this case
    "indoor_imu_only": dict(kind="indoor", n_sweeps=20, W=6, Wo=3, iwf=1, io=2, cfg=dict(point_distance_factor=0), prior_layout_differs=True),
    # SURVEY.md 8(d) config 4's stress setting: the whole window optimised (Wo = W = 15) on the HDL-64E
    "outdoor64_15_15": dict(kind="outdoor", n_sweeps=57, W=15, Wo=15, iwf=1, io=3, cfg={}),
    # estimate_extrinsic = 2: the hand-eye rotation has to converge first (it does not on this motion: same refusals on both sides)
    "indoor_extrinsic2": dict(kind="indoor", n_sweeps=18, W=6, Wo=3, iwf=1, io=2, cfg=dict(extrinsic_stage=2)),
}


_SWEEPS = {}


def sweeps_of(kind, n):
    """synth.make_sweeps(kind, n), generated once per process for the longest n asked for (sweep k does not depend on n; ray casting
    an HDL-64E sweep takes about a second)"""
    from lio_amd import synth

    have = _SWEEPS.get(kind)
    if have is None or len(have[0]) < n:
        _SWEEPS[kind] = have = synth.make_sweeps(kind, max(n, max(c["n_sweeps"] for c in CASES.values() if c["kind"] == kind and not c.get("fixture"))))
    return have


def feature_digest(pt, co):
    """order-free summary of one frame's plane factors"""
    return np.concatenate([[len(pt)], pt.sum(axis=0) if len(pt) else np.zeros(3), co.sum(axis=0) if len(co) else np.zeros(4),
                           [np.abs(co[:, 3]).sum() if len(co) else 0.0]])


def run_case(lib, name, est_factory=None, features_of=None, force_from=None):
    """Replays case `name`.  lib: the oracle library (front end; also the estimator unless est_factory(cfg) supplies another).
    features_of(est, opt_frame) -> (points, coeffs) of one opt-window frame of the last solve.  force_from: rows of an earlier run
    (the reference's); after every message of an initialised estimator its window, extrinsic and marginalization prior are
    overwritten with that run's (teacher forcing: every step then starts from the same state on both sides).
    Returns one dict of arrays per processed message."""
    c = CASES[name]
    W, Wo = c["W"], c["Wo"]
    rows = []

    def configure(cfg):
        for k, v in c["cfg"].items():
            setattr(cfg, k, v)

    def on_step(rp, k, e):
        est = rp.est
        st = est.stage()
        T = e["T_to_init"]
        r = dict(event=e["event"], T=np.concatenate([np.asarray(T[0], float), np.asarray(T[1], float)]), inited=st["inited"],
                 extrinsic_stage=st["extrinsic_stage"], cir_buf_count=st["cir_buf_count"])
        if st["inited"]:
            rep = e["report"]
            w = est.get_window()
            r.update(Ps=w["Ps"], Rs=w["Rs"], Vs=w["Vs"], Bas=w["Bas"], Bgs=w["Bgs"], lb=np.concatenate([w["q_lb"], w["t_lb"]]).astype(float),
                     g_vec=st["g_vec"], R_WI=st["R_WI"], iterations=rep.iterations, termination=rep.termination,
                     n_lidar=rep.n_lidar_residuals, initial_cost=rep.initial_cost, final_cost=rep.final_cost,
                     trace=np.asarray(rep.cost_trace[:11], float))
            r["feats"] = np.stack([feature_digest(*features_of(est, i)) for i in range(1, Wo + 1)])
            lm = est.local_map()
            r["local_map"] = np.concatenate([[lm.shape[0]], lm[:, :3].astype(float).sum(axis=0) if len(lm) else np.zeros(3)])
            pr = est.prior()
            if pr is not None:
                r.update(prior_n=pr["n"], JtJ=pr["JtJ"], Jtr=pr["Jtr"], x0=pr["x0"])
                if "lin_jac" in pr:
                    r.update(prior_jac=pr["lin_jac"], prior_res=pr["lin_res"])
            if force_from is not None:
                f = force_from[len(rows)]
                est.set_window(f["Ps"], f["Rs"], f["Vs"], f["Bas"], f["Bgs"], f["g_vec"])
                est.set_extrinsic(f["lb"][:4], f["lb"][4:])
                if "prior_jac" in f and pr is not None:
                    est.set_prior_factor(dict(n=int(f["prior_n"]), lin_jac=f["prior_jac"], lin_res=f["prior_res"], x0=f["x0"]))
        rows.append(r)

    extra = {}
    if c.get("fixture"):
        from fixture_util import fixture_sweeps

        if "fixture" not in _SWEEPS:
            _SWEEPS["fixture"] = fixture_sweeps(max(k["n_sweeps"] for k in CASES.values() if k.get("fixture")))
        sweeps, traj = _SWEEPS["fixture"]
        extra = dict(sweeps=sweeps, traj=traj, t0=0.0)
    else:
        extra = dict(sweeps=sweeps_of(c["kind"], c["n_sweeps"]))
    run_from_zero(lib, c["n_sweeps"], W=W, Wo=Wo, init_window_factor=c["iwf"], odom_io=c["io"], kind=c["kind"], configure=configure,
                  on_step=on_step, est_factory=est_factory, **extra)
    return rows


def oracle_features(W, Wo):
    piv = W - Wo
    return lambda est, i: est.features(piv + i)[:2]


def ref_features(est, i):
    return est.features(i)


def pack(rows):
    """rows -> flat dict of arrays (npz)"""
    out = {"events": np.array([r["event"] for r in rows])}
    for k, r in enumerate(rows):
        for key, v in r.items():
            if key != "event":
                out["%03d/%s" % (k, key)] = np.asarray(v)
    return out


def unpack(npz, prefix):
    ev = [str(e) for e in npz[prefix + "/events"]]
    rows = [dict(event=e) for e in ev]
    for key in npz.files:
        if not key.startswith(prefix + "/") or key.endswith("/events"):
            continue
        _, k, field = key.split("/")
        rows[int(k)][field] = npz[key]
    return rows
