"""The COMPILED drop-in: lio-mapping_amd/dropin/EstimatorHip.{h,cc} — a class with lio::Estimator's public surface over
include/lio_c.h — built against the reference's headers with the reference's OWN MeasurementManager.cc underneath it
(oracle/dropin_harness.cc -> oracle/_ref/libdropin_estimator.so, `make -C oracle ref`), driven the way estimator_node.cc:142-153 drives
lio::Estimator: construct, SetupRos, ProcessEstimation on its own thread, ImuHandler / CompactDataHandler as the subscriber callbacks.

A replay from t = 0 (case `indoor` of tests/ref_est_cases.py; the front end is the oracle's, as in the golden file) is fed to it message by
message, and after every processed /compact_data message
  * the state the class mirrors under the reference's member names (stage_flag_, cir_buf_count_, Ps_ / Rs_ / Vs_ / Bas_ / Bgs_ by
    CircularBuffer index, transform_lb_, transform_aft_mapped_, R_WI_, g_vec_) equals, BIT FOR BIT, that of the same library driven through
    lio_amd.replay (the Python mirror of GetMeasurements + ProcessEstimation that every other replay test uses) — i.e. the reference's
    pairing code and the class's IMU interpolation feed the library exactly what the tested path feeds it;
  * /predict_laser_odom and /local_laser_odom carry the lidar poses of Estimator.cc:728-758 computed from that window;
  * events, window and T_to_init agree with what the REFERENCE's Estimator.cc produced on these messages
    (tests/golden/ref_estimator_run.npz) at the free-running bounds of tests/test_gpu_ref_estimator.py (the step-by-step statement at
    1e-4 is tests/test_gpu_ref_estimator_steps.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import ref_est_cases as cases
from lio_amd import capi
from replay_util import run_from_zero

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "..", "oracle", "_ref", "libdropin_estimator.so")
GOLDEN = os.path.join(HERE, "golden", "ref_estimator_run.npz")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _load():
    assert os.path.exists(SO), "oracle/_ref/libdropin_estimator.so is missing: run `make -C oracle ref` where /root/reference exists (build())"
    lib = C.CDLL(SO)
    lib.dropin_create.restype = C.c_void_p
    lib.dropin_create.argtypes = [C.c_void_p] * 3 + [C.c_double] * 2
    lib.dropin_destroy.argtypes = [C.c_void_p]
    lib.dropin_push_imu.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    lib.dropin_push_compact.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_size_t]
    lib.dropin_wait_processed.argtypes = [C.c_void_p, C.c_size_t, C.c_double]
    lib.dropin_wait_processed.restype = C.c_int
    lib.dropin_get_stage.argtypes = [C.c_void_p] * 4
    lib.dropin_get_window.argtypes = [C.c_void_p] * 8
    lib.dropin_get_window.restype = C.c_int
    lib.dropin_get_published.argtypes = [C.c_void_p] * 4
    return lib


def _create(lib, cfg):
    ip = np.array([cfg.window_size, cfg.opt_window_size, cfg.init_window_factor, cfg.extrinsic_stage, cfg.opt_extrinsic, cfg.imu_factor,
                   cfg.point_distance_factor, cfg.prior_factor, cfg.marginalization_factor, cfg.enable_deskew, cfg.cutoff_deskew, cfg.keep_features], np.int32)
    fp = np.array([cfg.corner_filter_size, cfg.surf_filter_size, cfg.min_match_sq_dis, cfg.min_plane_dis] + list(cfg.transform_lb.q) + list(cfg.transform_lb.p), np.float32)
    dp = np.array([cfg.acc_n, cfg.gyr_n, cfg.acc_w, cfg.gyr_w, cfg.g_norm], np.float64)
    h = lib.dropin_create(_p(ip), _p(fp), _p(dp), 0.0, float(cfg.max_solver_time))
    assert h, "EstimatorHip could not create its library handle (no GPU?)"
    return h


def test_dropin_class_replays_like_the_c_abi_and_the_reference(hip, oracle):
    name = "indoor"
    c = cases.CASES[name]
    gold = cases.unpack(np.load(GOLDEN), name)
    lib = _load()
    W = c["W"]
    n = W + 1
    state = dict(h=None, worst_T=0.0, worst={}, solved=0)

    def tap(cfg, kind, *m):
        if state["h"] is None:
            state["h"] = _create(lib, cfg)
        if kind == "imu":
            t, acc, gyr = m
            lib.dropin_push_imu(state["h"], t, _p(np.ascontiguousarray(acc, np.float64)), _p(np.ascontiguousarray(gyr, np.float64)))
        else:
            stamp, compact = m
            cl = np.ascontiguousarray(compact, np.float32).reshape(-1, 4)
            lib.dropin_push_compact(state["h"], stamp, _p(cl), cl.shape[0])

    def on_step(rp, k, e):
        h = state["h"]
        assert lib.dropin_wait_processed(h, len(rp.log), 60.0) == 0, "thread B did not finish the message the Python replay has paired"
        o5, R, g = np.zeros(5, np.int32), np.zeros((3, 3)), np.zeros(3)
        lib.dropin_get_stage(h, _p(o5), _p(R), _p(g))
        st = rp.est.stage()
        assert o5[4] == 0, f"library call failed inside the drop-in: code {o5[4]}"
        assert bool(o5[0]) == st["inited"] and o5[1] == st["cir_buf_count"] and o5[2] == st["extrinsic_stage"]
        assert capi.Estimator.EVENTS[o5[3]] == st["event"] == str(gold[len(rp.log) - 1]["event"])
        Ps, Rs, Vs, Bas, Bgs = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        lb, aft = np.zeros(7, np.float32), np.zeros(7, np.float32)
        held = lib.dropin_get_window(h, _p(Ps), _p(Rs), _p(Vs), _p(Bas), _p(Bgs), _p(lb), _p(aft))
        w = rp.est.get_window()
        assert held == (n if st["inited"] else min(n, st["cir_buf_count"] + 1))
        for key, got in (("Ps", Ps), ("Rs", Rs), ("Vs", Vs), ("Bas", Bas), ("Bgs", Bgs)):
            np.testing.assert_array_equal(got[:held], w[key][:held], err_msg=key)       # same library, same inputs: the same bits
        np.testing.assert_array_equal(lb, np.concatenate([w["q_lb"], w["t_lb"]]).astype(np.float32))
        np.testing.assert_array_equal(aft, np.concatenate([np.asarray(e["T_to_init"][0], np.float32), np.asarray(e["T_to_init"][1], np.float32)]))
        np.testing.assert_array_equal(R, st["R_WI"])
        np.testing.assert_array_equal(g, st["g_vec"])
        r = gold[len(rp.log) - 1]
        state["worst_T"] = max(state["worst_T"], float(np.abs(aft[4:] - r["T"][4:]).max()))
        if st["inited"]:
            state["solved"] += 1
            laser, local, rep3 = np.zeros(9), np.zeros(9), np.zeros(3)
            lib.dropin_get_published(h, _p(laser), _p(local), _p(rep3))
            assert int(rep3[0]) == e["report"].iterations and int(rep3[1]) == e["report"].n_lidar_residuals and rep3[2] == e["report"].final_cost
            # /extrinsic_lb goes out with every solve; the two odometry topics only from the INITED branch (Estimator.cc:728-758), i.e.
            # not on the initialising step (:541-600 runs SolveOptimization + SlideWindow and publishes nothing else)
            if e["event"] != "initialised":
                state["published"] = state.get("published", 0) + 1
                assert laser[0] == e["stamp"]
            assert int(laser[1]) == state.get("published", 0)
            if e["event"] == "initialised":
                return
            # /predict_laser_odom = the newest frame's LIDAR pose (Estimator.cc:744-758): R_last * R_lb^-1, P_last - that * t_lb
            qlb, tlb = w["q_lb"].astype(float), w["t_lb"].astype(float)
            x, y, z, s = qlb
            Rlb = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * s), 2 * (x * z + y * s)], [2 * (x * y + z * s), 1 - 2 * (x * x + z * z), 2 * (y * z - x * s)],
                            [2 * (x * z - y * s), 2 * (y * z + x * s), 1 - 2 * (x * x + y * y)]])
            for msg, idx in ((laser, W), (local, W - c["Wo"])):
                Rl = w["Rs"][idx] @ Rlb.T
                np.testing.assert_allclose(msg[6:9], w["Ps"][idx] - Rl @ tlb, atol=1e-6)   # (the fp32 extrinsic quaternion is a unit quaternion to 1e-7)
                qx, qy, qz, qw = msg[2:6]
                Rq = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)], [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                               [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
                np.testing.assert_allclose(Rq, Rl, atol=1e-6)
            for key, got in (("Ps", Ps), ("Rs", Rs), ("Vs", Vs), ("Bgs", Bgs)):
                state["worst"][key] = max(state["worst"].get(key, 0.0), float(np.abs(got - r[key]).max()))

    n_sweeps = 26                                   # 13 laser messages: the stretch tests/test_gpu_ref_estimator.py covers
    try:
        rp, _ = run_from_zero(oracle, n_sweeps, W=W, Wo=c["Wo"], init_window_factor=c["iwf"], odom_io=c["io"], on_step=on_step, tap=tap,
                              est_factory=lambda cfg: capi.Estimator(hip, cfg), sweeps=cases.sweeps_of("indoor", n_sweeps))
        ev = [e["event"] for e in rp.log]
        assert len(ev) >= 12 and ev == [str(r["event"]) for r in gold[:len(ev)]] and "initialised" in ev and state["solved"] >= 5
        print("drop-in class == C-ABI replay bit for bit over", len(ev), "messages,", state["solved"], "solves; vs the reference's Estimator.cc: worst T_to_init diff",
              state["worst_T"], "worst window diffs", state["worst"])
        assert state["worst_T"] < 0.03
        assert state["worst"]["Ps"] < 0.15 and state["worst"]["Rs"] < 0.01 and state["worst"]["Vs"] < 0.15 and state["worst"]["Bgs"] < 3e-3, state["worst"]
    finally:
        if state["h"]:
            lib.dropin_destroy(state["h"])
