"""ctypes adapter over oracle/_ref/libref_estimator.so — the REFERENCE's own Estimator compiled against the stand-in headers of
oracle/ref_shim (see oracle/ref_estimator.cc for what runs and what is stood in).  It offers the few methods of lio_amd.capi.Estimator
that lio_amd.replay.Replay calls, so that the same replay can drive either.  Build container only (/root/reference is needed to
build the library); the committed digests in tests/golden/ref_estimator_run.npz are what travels."""
import ctypes as C
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "oracle", "_ref", "libref_estimator.so")
EVENTS = ("skipped", "filling", "init_failed", "initialised", "solved")


def load():
    lib = C.CDLL(LIB)
    lib.ref_est_create.restype = C.c_void_p
    lib.ref_est_create.argtypes = [C.c_void_p] * 3
    lib.ref_est_destroy.argtypes = [C.c_void_p]
    lib.ref_est_process_imu.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_double]
    lib.ref_est_process_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p]
    lib.ref_est_process_compact.restype = C.c_int
    lib.ref_est_get_stage.argtypes = [C.c_void_p] * 7
    lib.ref_est_get_window.argtypes = [C.c_void_p] * 7
    lib.ref_est_get_features.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.ref_est_get_features.restype = C.c_size_t
    lib.ref_est_get_local_map.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_est_get_local_map.restype = C.c_size_t
    lib.ref_est_get_surf_stack.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ref_est_get_surf_stack.restype = C.c_size_t
    lib.ref_est_get_prior.argtypes = [C.c_void_p] * 9 + [C.c_int]
    lib.ref_est_get_prior.restype = C.c_int
    lib.ref_est_get_solve_params.argtypes = [C.c_void_p] * 5
    lib.ref_est_get_para.argtypes = [C.c_void_p] * 2
    lib.ref_est_get_imu_factor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.ref_est_get_imu_factor.restype = C.c_int
    lib.ref_est_get_preintegration.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.ref_est_get_preintegration.restype = C.c_int
    lib.ref_est_get_tmp_preintegration.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_mm_create.restype = C.c_void_p
    lib.ref_mm_create.argtypes = [C.c_double]
    lib.ref_mm_destroy.argtypes = [C.c_void_p]
    lib.ref_mm_push_imu.argtypes = [C.c_void_p, C.c_double]
    lib.ref_mm_push_compact.argtypes = [C.c_void_p, C.c_double]
    lib.ref_mm_get_measurements.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.ref_mm_get_measurements.restype = C.c_int
    return lib


def mm_pairings(lib, delay, msgs):
    """rows (index of the message after which the pairing came out, laser stamp, number of IMU messages, first and last IMU stamp)"""
    h = lib.ref_mm_create(float(delay))
    rows, buf = [], np.zeros(4 * 64)
    for k, (kind, stamp) in enumerate(msgs):
        (lib.ref_mm_push_imu if kind == "imu" else lib.ref_mm_push_compact)(h, float(stamp))
        n = lib.ref_mm_get_measurements(h, _p(buf), 64)
        for j in range(n):
            rows.append([k] + list(buf[4 * j:4 * j + 4]))
    lib.ref_mm_destroy(h)
    return rows


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefEstimator:
    def __init__(self, lib, cfg):
        """cfg: a lio_amd.capi.EstConfig (the fields the reference's EstimatorConfig has are taken over)"""
        self.lib, self.cfg = lib, cfg
        self.W, self.Wo = cfg.window_size, cfg.opt_window_size
        ip = np.array([cfg.window_size, cfg.opt_window_size, cfg.init_window_factor, cfg.extrinsic_stage, cfg.opt_extrinsic, cfg.imu_factor,
                       cfg.point_distance_factor, cfg.prior_factor, cfg.marginalization_factor, cfg.enable_deskew, cfg.cutoff_deskew,
                       cfg.keep_features], np.int32)
        q, p = list(cfg.transform_lb.q), list(cfg.transform_lb.p)
        fp = np.array([cfg.corner_filter_size, cfg.surf_filter_size, cfg.min_match_sq_dis, cfg.min_plane_dis] + q + p, np.float32)
        dp = np.array([cfg.acc_n, cfg.gyr_n, cfg.acc_w, cfg.gyr_w, cfg.g_norm], np.float64)
        assert cfg.max_num_iterations == 10          # hard-coded in the reference (Estimator.cc:1916)
        self.h = lib.ref_est_create(_p(ip), _p(fp), _p(dp))
        self.event = "skipped"

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_est_destroy(self.h)
            self.h = None

    def process_imu(self, dt, acc, gyr, stamp):
        a, g = np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(gyr, np.float64)
        self.lib.ref_est_process_imu(self.h, float(dt), _p(a), _p(g), float(stamp))

    def process_compact(self, compact, stamp):
        c = np.ascontiguousarray(compact, np.float32).reshape(-1, 4)
        T, rep = np.zeros(7, np.float32), np.zeros(48, np.float64)
        ev = self.lib.ref_est_process_compact(self.h, _p(c), c.shape[0], float(stamp), _p(T), _p(rep))
        self.event = EVENTS[ev]
        n_eval = int(rep[7])
        costs = list(rep[8:8 + n_eval])
        r = types.SimpleNamespace(iterations=int(rep[0]), successful_steps=int(rep[1]), termination=int(rep[2]), n_lidar_residuals=int(rep[3]),
                                  n_blocks=int(rep[4]), initial_cost=float(rep[5]), final_cost=float(rep[6]), evaluate_costs=costs,
                                  cost_trace=list(rep[16:48]))
        return (T[:4].copy(), T[4:].copy()), r

    def stage(self):
        st, cb, ex, cv = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        R, g = np.zeros((3, 3)), np.zeros(3)
        self.lib.ref_est_get_stage(self.h, C.byref(st), C.byref(cb), C.byref(ex), C.byref(cv), _p(R), _p(g))
        return dict(inited=bool(st.value), cir_buf_count=cb.value, extrinsic_stage=ex.value, convergence=bool(cv.value), event=self.event, R_WI=R, g_vec=g)

    def get_window(self):
        n = self.W + 1
        Ps, Rs, Vs, Bas, Bgs = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        lb = np.zeros(7, np.float32)
        self.lib.ref_est_get_window(self.h, _p(Ps), _p(Rs), _p(Vs), _p(Bas), _p(Bgs), _p(lb))
        return dict(Ps=Ps, Rs=Rs, Vs=Vs, Bas=Bas, Bgs=Bgs, q_lb=lb[:4].copy(), t_lb=lb[4:].copy())

    def features(self, frame):
        n = self.lib.ref_est_get_features(self.h, frame, None, None)
        pt, co = np.zeros((n, 3)), np.zeros((n, 4))
        if n:
            self.lib.ref_est_get_features(self.h, frame, _p(pt), _p(co))
        return pt, co

    def local_map(self):
        n = self.lib.ref_est_get_local_map(self.h, None)
        out = np.zeros((n, 4), np.float32)
        if n:
            self.lib.ref_est_get_local_map(self.h, _p(out))
        return out

    def surf_stack(self, frame):
        n = self.lib.ref_est_get_surf_stack(self.h, frame, None)
        out = np.zeros((n, 4), np.float32)
        if n:
            self.lib.ref_est_get_surf_stack(self.h, frame, _p(out))
        return out

    def solve_problem(self):
        """the problem of the last solve as the reference handed it to ceres::Solve: start / end parameters ((Wo + 1) x (7 + 9) + 7),
        flags, the extrinsic prior's constants, the parameter arrays as they stand now (the marginalization's linearisation point),
        and per interval the raw IMU samples (None where the reference added no ImuFactor)"""
        n = (self.Wo + 1) * 16 + 7
        ini, fin, para, flags, pr7 = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(3, np.int32), np.zeros(7)
        self.lib.ref_est_get_solve_params(self.h, _p(ini), _p(fin), _p(flags), _p(pr7))
        self.lib.ref_est_get_para(self.h, _p(para))
        imu = []
        for i in range(self.Wo):
            head, smp = np.zeros(12), np.zeros(7 * 4096)
            k = self.lib.ref_est_get_imu_factor(self.h, i, _p(head), _p(smp), 4096)
            assert k >= -1
            imu.append(None if k < 0 else (head, smp[:7 * k].reshape(k, 7).copy()))
        return dict(initial=ini, final=fin, para=para, ex_constant=int(flags[0]), has_prior=int(flags[1]), use_prior_factor=int(flags[2]), prior7=pr7, imu=imu)

    def state_dump(self):
        """everything another implementation needs to continue from where this estimator stands (after a processed message): the
        window, extrinsic, gravity, per window slot the surf stack and the pre-integration's raw samples, what the pre-integration in
        flight was started from, the prior (canonical order)"""
        w, st = self.get_window(), self.stage()
        d = dict(Ps=w["Ps"], Rs=w["Rs"], Vs=w["Vs"], Bas=w["Bas"], Bgs=w["Bgs"], lb=np.concatenate([w["q_lb"], w["t_lb"]]).astype(float), g_vec=st["g_vec"])
        for i in range(self.W + 1):
            d["stack%d" % i] = self.surf_stack(i)
            head, smp = np.zeros(12), np.zeros(7 * 4096)
            k = self.lib.ref_est_get_preintegration(self.h, i, _p(head), _p(smp), 4096)
            assert k >= -1
            if k >= 0:
                d["pre%d_head" % i], d["pre%d_samples" % i] = head, smp[:7 * k].reshape(k, 7).copy()
        tmp = np.zeros(12)
        self.lib.ref_est_get_tmp_preintegration(self.h, _p(tmp))
        d["tmp_head"] = tmp
        pr = self.prior()
        if pr is not None:
            d.update(prior_n=np.array(pr["n"]), prior_jac=pr["lin_jac"], prior_res=pr["lin_res"], prior_x0=pr["x0"])
        return d

    def prior(self):
        """dict(n, JtJ, Jtr, x0) in the oracle's canonical kept order (pose1, sb1, pose2 .. poseWo, extrinsic -> after the address shift:
        pose 0, speed-bias 0, pose 1 .. pose Wo-1, extrinsic), or None"""
        cap = 15 * (self.Wo + 2)
        J, r = np.zeros((cap, cap)), np.zeros(cap)
        nb = C.c_int()
        kind, index, offset, size = (np.zeros(64, np.int32) for _ in range(4))
        x0 = np.zeros(16 * 64)
        n = self.lib.ref_est_get_prior(self.h, _p(J), _p(r), C.byref(nb), _p(kind), _p(index), _p(offset), _p(size), _p(x0), cap)
        if n <= 0:
            return None
        J = J.reshape(-1)[:n * n].reshape(n, n)
        r = r[:n]
        blocks, xo = [], 0
        for k in range(nb.value):
            blocks.append((int(kind[k]), int(index[k]), int(offset[k]), int(size[k]), x0[xo:xo + size[k]].copy()))
            xo += int(size[k])
        order = sorted(range(len(blocks)), key=lambda k: (blocks[k][0] == 2, blocks[k][1], blocks[k][0]))
        cols, xs = [], []
        for k in order:
            kd, ix, off, sz, x = blocks[k]
            loc = 6 if sz == 7 else sz
            cols += list(range(off, off + loc))
            xs.append(x)
        Jc = J[:, cols]
        return dict(n=n, JtJ=Jc.T @ Jc, Jtr=Jc.T @ r, x0=np.concatenate(xs), blocks=[blocks[k][:4] for k in order], lin_jac=Jc, lin_res=r.copy())
