"""One estimator step taken from THE REFERENCE'S state (shared by tests/test_ref_estimator_state.py — oracle, CPU — and
tests/test_gpu_ref_estimator.py — product, GPU).

tests/golden/ref_estimator_states.npz holds what the reference's own Estimator.cc (oracle/ref_estimator.cc) had in its buffers after
a laser message of three replays of tests/ref_est_cases.py — window states, extrinsic, gravity, the surf stack and
the raw IMU samples of the pre-integration of every window slot, what the pre-integration in flight was started from, the
marginalization prior —, the next message as it was fed (IMU batch, /compact_data) and what the reference had after it.  `one_step(lib)` builds an estimator of
`lib`, injects the state through the test hooks of the C-ABI (lio_est_set_window / set_surf_stack / set_preintegration /
begin_frame / set_extrinsic), feeds the next message once (so that the estimator owns a prior of the right shape), overwrites EVERYTHING
with the reference's state again (now including the prior), feeds the same message, and returns what came out: one
ProcessImu ... ProcessLaserOdom -> SolveOptimization -> SlideWindow step from exactly the reference's state, to be compared with the
reference's own next state.  Index conventions are the reference's: buffers are dumped by the logical index of its CircularBuffers after
SlideWindow (states already shifted, stacks and pre-integrations shifted by the next push), which is also how the oracle and the product
hold them."""
import os

import numpy as np

import ref_est_cases as cases
from lio_amd import capi, pipeline

HERE = os.path.dirname(os.path.abspath(__file__))
STATES = os.path.join(HERE, "golden", "ref_estimator_states.npz")
RUN = os.path.join(HERE, "golden", "ref_estimator_run.npz")
# case of tests/ref_est_cases.py -> the estimator step after the initialisation (0 = the step that initialised) whose state is stored;
# compared: the step after it.  `indoor`: VLP-16, kept features, IMU de-skew, free extrinsic, no prior factor; `outdoor64`: HDL-64E,
# extrinsic PriorFactor, cut-off de-skew; `outdoor64_15_5`: the same at BASELINE.json's headline window (15 / 5, every third sweep)
STEPS = {"indoor": 4, "outdoor64": 2, "outdoor64_15_5": 2}


def config_for(lib, CASE):
    c = cases.CASES[CASE]
    if c["kind"] == "indoor":                                    # (what run_from_zero builds for the two kinds)
        cfg = pipeline.config_indoor(lib, c["W"], c["Wo"])
        cfg.transform_lb = capi.TransformF.make([0, 0, 0, 1], [0.0, 0.0, -0.081939])
    else:
        cfg = pipeline.config_outdoor64(lib, c["W"], c["Wo"])
        cfg.transform_lb = capi.TransformF.make([0, 0, 0, 1], [-8.086759e-01, 3.195559e-01, -7.997231e-01])
    cfg.init_window_factor, cfg.extrinsic_stage = c["iwf"], 1
    for k, v in c["cfg"].items():
        setattr(cfg, k, v)
    return cfg


def inject(est, d, with_prior):
    W = est.W
    # the pre-integration in flight keeps the biases it was started with (the newest frame's BEFORE the solve): hand those to begin_frame
    # through slot W, then set the real window
    Bas, Bgs = np.array(d["Bas"]), np.array(d["Bgs"])
    Bas[W], Bgs[W] = d["tmp_head"][6:9], d["tmp_head"][9:12]
    est.set_window(d["Ps"], d["Rs"], d["Vs"], Bas, Bgs, d["g_vec"])
    est.begin_frame(d["tmp_head"][0:3], d["tmp_head"][3:6])
    est.set_window(d["Ps"], d["Rs"], d["Vs"], d["Bas"], d["Bgs"], d["g_vec"])
    est.set_extrinsic(d["lb"][:4], d["lb"][4:])
    for i in range(W + 1):
        # slots that lie behind the pivot after the next push are not stored (spent local maps, never read again by a solve): one point far
        # from the scene stands in
        est.set_surf_stack(i, d["stack%d" % i] if "stack%d" % i in d else np.array([[200.0, 200.0, 50.0, 0.0]], np.float32))
        if "pre%d_head" % i in d:
            h, s = d["pre%d_head" % i], d["pre%d_samples" % i]
            est.set_preintegration(i, h[0:3], h[3:6], h[6:9], h[9:12], s[:, 0], s[:, 1:4], s[:, 4:7])
    if with_prior:
        est.set_prior_factor(dict(n=int(d["prior_n"]), lin_jac=d["prior_jac"], lin_res=d["prior_res"], x0=d["prior_x0"]))


def feed(est, msg):
    """one laser message after the initialisation: ProcessImu per sample, then ProcessLaserOdom on the message's clouds — what
    lio_est_process_compact does then (the transform it hands over only goes into the initialisation's buffers), through the entry point
    the injected-window tests use (lio_amd.pipeline.feed_frame)"""
    batch, compact, stamp = msg
    for dt, acc, gyr, t in batch:
        est.process_imu(dt, acc, gyr, t)
    _, corner, surf, _ = est.lib.compact_decode(compact)
    return None, est.process_laser_odom(capi.TransformF.make([0, 0, 0, 1], [0, 0, 0]), surf, corner, stamp)


def load_states(CASE):
    """-> B (the injected state), C (what the reference had one message later), M (that message: IMU batch, /compact_data, stamp)"""
    g = np.load(STATES)
    out = {}
    for key in g.files:
        case, tag, field = key.split("/")
        if case == CASE:
            out.setdefault(tag, {})[field] = g[key]
    m = out["M"]
    msg = ([(float(r[0]), r[1:4].copy(), r[4:7].copy(), float(r[7])) for r in m["imu"]], m["compact"], float(m["stamp"]))
    return out["B"], out["C"], msg


def one_step(lib, CASE="indoor"):
    """-> `lib`'s estimator after ONE step from the reference's state B, its solve report, and what the reference itself had after that
    step (C: window, extrinsic, prior, solve summary, local-map digest — from the same run of the reference as B and the message)"""
    B, C, msg = load_states(CASE)
    est = capi.Estimator(lib, config_for(lib, CASE))
    inject(est, B, with_prior=False)
    feed(est, msg)                 # (only so that the estimator owns a prior of the right shape: everything it did is overwritten next)
    inject(est, B, with_prior=True)
    _, rep = feed(est, msg)
    return est, rep, C
