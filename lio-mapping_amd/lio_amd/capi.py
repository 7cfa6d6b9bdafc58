"""ctypes view of include/lio_c.h.

`load_hip()` opens the product (liblio_hip.so next to csrc/) and raises if it is missing: there is no
CPU fallback.  Tests additionally open the oracle with `LioLib(path_to_liblio_oracle)`; product code
never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("LIO_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "liblio_hip.so")   # LIO_HIP_LIB: A/B builds of the product

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_uint16_p = C.POINTER(C.c_uint16)


class TransformF(C.Structure):
    _fields_ = [("q", C.c_float * 4), ("p", C.c_float * 3)]

    @staticmethod
    def make(q_xyzw, p):
        t = TransformF()
        for i in range(4):
            t.q[i] = float(q_xyzw[i])
        for i in range(3):
            t.p[i] = float(p[i])
        return t

    def to_np(self):
        return np.array(list(self.q), dtype=np.float32), np.array(list(self.p), dtype=np.float32)


class PPConfig(C.Structure):
    _fields_ = [
        ("scan_period", C.c_double),
        ("num_scan_subregions", C.c_int),
        ("num_curvature_regions", C.c_int),
        ("surf_curv_th", C.c_float),
        ("max_corner_sharp", C.c_int),
        ("max_corner_less_sharp", C.c_int),
        ("max_surf_flat", C.c_int),
        ("less_flat_filter_size", C.c_float),
        ("infer_start_ori", C.c_int),
        ("rad_diff", C.c_double),
    ]


class MapConfig(C.Structure):
    _fields_ = [
        ("corner_filter_size", C.c_float),
        ("surf_filter_size", C.c_float),
        ("min_match_sq_dis", C.c_float),
        ("min_plane_dis", C.c_float),
        ("num_max_iterations", C.c_int),
        ("map_builder", C.c_int),
        ("enable_4d", C.c_int),
        ("skip_count", C.c_int),
    ]


class EstConfig(C.Structure):
    _fields_ = [
        ("window_size", C.c_int),
        ("opt_window_size", C.c_int),
        ("corner_filter_size", C.c_float),
        ("surf_filter_size", C.c_float),
        ("min_match_sq_dis", C.c_float),
        ("min_plane_dis", C.c_float),
        ("transform_lb", TransformF),
        ("opt_extrinsic", C.c_int),
        ("imu_factor", C.c_int),
        ("point_distance_factor", C.c_int),
        ("prior_factor", C.c_int),
        ("marginalization_factor", C.c_int),
        ("enable_deskew", C.c_int),
        ("cutoff_deskew", C.c_int),
        ("keep_features", C.c_int),
        ("acc_n", C.c_double),
        ("gyr_n", C.c_double),
        ("acc_w", C.c_double),
        ("gyr_w", C.c_double),
        ("g_norm", C.c_double),
        ("max_num_iterations", C.c_int),
        ("max_solver_time", C.c_double),
        ("extrinsic_stage", C.c_int),
        ("init_window_factor", C.c_int),
        ("device_solve", C.c_int),
        ("inline_marg", C.c_int),
        ("stream_sync", C.c_int),
        ("moments_form", C.c_int),
        ("resident_moments", C.c_int),
    ]


class SolveReport(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("successful_steps", C.c_int),
        ("termination", C.c_int),
        ("n_lidar_residuals", C.c_int),
        ("n_local_map", C.c_int),
        ("laser_odom_iterations", C.c_int),
        ("turn_off", C.c_int),
        ("convergence_flag", C.c_int),
        ("marginalized", C.c_int),
        ("cost_pim_before", C.c_double),
        ("cost_ppp_before", C.c_double),
        ("cost_marg_before", C.c_double),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("cost_trace", C.c_double * 32),
        ("ms_build_map", C.c_double),
        ("ms_features", C.c_double),
        ("ms_prepare", C.c_double),
        ("ms_opt", C.c_double),
        ("ms_marg", C.c_double),
        ("ms_total", C.c_double),
        ("laser_odom_kz", C.c_int),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if name == "cost_trace" else v
        return d


# every symbol include/lio_c.h declares (checked by tests/test_abi.py against the header text)
_SIGS = {
    "lio_backend": (C.c_char_p, []),
    "lio_pp_default_config": (None, [C.POINTER(PPConfig)]),
    "lio_pp_create": (C.c_void_p, [C.c_float, C.c_float, C.c_int, C.POINTER(PPConfig)]),
    "lio_pp_destroy": (None, [C.c_void_p]),
    "lio_pp_process": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t]),
    "lio_pp_process_rings": (C.c_int, [C.c_void_p, c_float_p, c_uint16_p, C.c_size_t]),
    "lio_bench_voxel_grid": (C.c_int, [c_float_p, C.c_size_t, C.c_float, C.c_int, c_double_p, C.POINTER(C.c_size_t)]),
    "lio_pp_process_async": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t]),
    "lio_pp_process_batch": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(c_float_p), C.POINTER(C.c_size_t), C.c_int]),
    "lio_pp_process_batch_device": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]),
    "lio_pp_process_rings_batch": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(c_float_p), C.POINTER(c_uint16_p), C.POINTER(C.c_size_t), C.c_int]),
    "lio_pp_wait": (C.c_int, [C.c_void_p]),
    "lio_pp_start_ori": (C.c_float, [C.c_void_p]),
    "lio_pp_count": (C.c_size_t, [C.c_void_p, C.c_int]),
    "lio_pp_get_cloud": (C.c_int, [C.c_void_p, C.c_int, c_float_p]),
    "lio_pp_get_indices": (C.c_int, [C.c_void_p, C.c_int, c_int32_p, c_int32_p]),
    "lio_pp_get_ring_offsets": (C.c_int, [C.c_void_p, c_int32_p]),
    "lio_pp_get_curvature": (C.c_int, [C.c_void_p, c_float_p, c_int32_p]),
    "lio_odom_create": (C.c_void_p, [C.c_float, C.c_int, C.c_int, C.c_int]),
    "lio_odom_destroy": (None, [C.c_void_p]),
    "lio_odom_process": (C.c_int, [C.c_void_p] + [c_float_p, C.c_size_t] * 4 + [C.POINTER(TransformF), C.POINTER(TransformF), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lio_odom_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "lio_odom_get_last_cloud": (C.c_size_t, [C.c_void_p, C.c_int, c_float_p]),
    "lio_odom_get_iteration_trace": (C.c_int, [C.c_void_p, C.POINTER(TransformF), C.c_int, C.POINTER(C.c_int)]),
    "lio_map_get_degeneracy": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "lio_kf_batch_get_degeneracy": (C.c_int, [C.c_void_p, c_int32_p]),
    "lio_imu_estimate_extrinsic_rotation": (C.c_int, [C.c_size_t, C.POINTER(TransformF), C.POINTER(C.c_void_p), C.POINTER(TransformF)]),
    "lio_imu_initialization": (C.c_int, [C.c_size_t, C.POINTER(TransformF), C.POINTER(C.c_void_p), C.POINTER(TransformF), c_double_p, c_double_p, c_double_p, c_double_p]),
    "lio_map_default_config": (None, [C.POINTER(MapConfig)]),
    "lio_map_create": (C.c_void_p, [C.POINTER(MapConfig)]),
    "lio_map_destroy": (None, [C.c_void_p]),
    "lio_map_process": (C.c_int, [C.c_void_p] + [c_float_p, C.c_size_t] * 2 + [C.POINTER(TransformF), C.POINTER(TransformF), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lio_map_set_init_flag": (C.c_int, [C.c_void_p, C.c_int]),
    "lio_map_set_transform_tobe_mapped": (C.c_int, [C.c_void_p, C.POINTER(TransformF)]),
    "lio_map_get_transform_tobe_mapped": (C.c_int, [C.c_void_p, C.POINTER(TransformF)]),
    "lio_map_update_map_database": (C.c_int, [C.c_void_p] + [c_float_p, C.c_size_t] * 2 + [C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(TransformF), C.POINTER(C.c_int)]),
    "lio_map_get_cloud": (C.c_size_t, [C.c_void_p, C.c_int, c_float_p]),
    "lio_map_get_cube": (C.c_size_t, [C.c_void_p, C.c_int, C.c_uint32, c_float_p]),
    "lio_map_get_cube_state": (C.c_size_t, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "lio_map_get_score_point_coeff": (C.c_size_t, [C.c_void_p, c_float_p, c_float_p, c_float_p]),
    "lio_kf_batch_create": (C.c_void_p, [C.POINTER(MapConfig)]),
    "lio_kf_batch_destroy": (None, [C.c_void_p]),
    "lio_kf_batch_add_map": (C.c_int, [C.c_void_p] + [c_float_p, C.c_size_t] * 2),
    "lio_kf_batch_add_keyframe": (C.c_int, [C.c_void_p, C.c_int] + [c_float_p, C.c_size_t] * 2 + [C.POINTER(TransformF)]),
    "lio_kf_batch_clear_keyframes": (C.c_int, [C.c_void_p]),
    "lio_kf_batch_refine": (C.c_int, [C.c_void_p, C.POINTER(TransformF), c_int32_p, c_int32_p, c_double_p]),
    "lio_kf_batch_size": (C.c_size_t, [C.c_void_p]),
    "lio_compact_encode": (C.c_size_t, [C.POINTER(TransformF)] + [c_float_p, C.c_size_t] * 3 + [c_float_p]),
    "lio_compact_decode": (C.c_int, [c_float_p, C.c_size_t, C.POINTER(TransformF)] + [C.POINTER(C.c_size_t)] * 3),
    "lio_voxel_grid": (C.c_int, [c_float_p, C.c_size_t, C.c_float, c_float_p, C.POINTER(C.c_size_t)]),
    "lio_knn": (C.c_int, [c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_int, C.c_float, c_int32_p, c_float_p]),
    "lio_calculate_features": (
        C.c_int,
        [c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.POINTER(TransformF), C.c_float, C.c_float, c_uint8_p, c_float_p, c_float_p],
    ),
    "lio_pim_create": (C.c_void_p, [c_double_p] * 4 + [C.c_double] * 5),
    "lio_pim_destroy": (None, [C.c_void_p]),
    "lio_pim_push_back": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_double_p]),
    "lio_pim_repropagate": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "lio_pim_get": (C.c_int, [C.c_void_p] + [c_double_p] * 6),
    "lio_pim_evaluate": (C.c_int, [C.c_void_p] + [c_double_p] * 5),
    "lio_factor_imu": (C.c_int, [C.c_void_p] + [c_double_p] * 9),
    "lio_factor_pivot_point_plane": (C.c_int, [c_double_p] * 9),
    "lio_factor_prior": (C.c_int, [c_double_p] * 5),
    "lio_pose_plus": (C.c_int, [c_double_p] * 3),
    "lio_est_default_config": (None, [C.POINTER(EstConfig)]),
    "lio_est_create": (C.c_void_p, [C.POINTER(EstConfig)]),
    "lio_est_destroy": (None, [C.c_void_p]),
    "lio_est_process_imu": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_double_p, C.c_double]),
    "lio_est_process_imu_batch": (C.c_int, [C.c_void_p, C.c_size_t, c_double_p, c_double_p, c_double_p, c_double_p]),
    "lio_est_process_laser_odom": (
        C.c_int,
        [C.c_void_p, C.POINTER(TransformF), c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_double, C.POINTER(SolveReport)],
    ),
    "lio_est_process_compact": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, C.c_double, C.POINTER(TransformF), C.POINTER(SolveReport)]),
    "lio_est_get_stage": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int)] * 4 + [c_double_p, c_double_p]),
    "lio_est_push_frame": (C.c_int, [C.c_void_p, C.POINTER(TransformF), c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_double]),
    "lio_est_solve_optimization": (C.c_int, [C.c_void_p, C.POINTER(SolveReport)]),
    "lio_est_slide_window": (C.c_int, [C.c_void_p]),
    "lio_est_sync": (C.c_int, [C.c_void_p]),
    "lio_est_set_window": (C.c_int, [C.c_void_p, C.c_int] + [c_double_p] * 6),
    "lio_est_get_window": (C.c_int, [C.c_void_p, C.c_int] + [c_double_p] * 5 + [C.POINTER(TransformF)]),
    "lio_est_set_surf_stack": (C.c_int, [C.c_void_p, C.c_int, c_float_p, C.c_size_t]),
    "lio_est_get_surf_stack": (C.c_size_t, [C.c_void_p, C.c_int, c_float_p]),
    "lio_est_set_preintegration": (C.c_int, [C.c_void_p, C.c_int] + [c_double_p] * 7 + [C.c_size_t]),
    "lio_est_begin_frame": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "lio_est_build_local_map": (C.c_int, [C.c_void_p]),
    "lio_est_get_local_map": (C.c_size_t, [C.c_void_p, c_float_p]),
    "lio_est_get_features": (C.c_size_t, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "lio_est_get_laser_odom_transform": (C.c_int, [C.c_void_p, C.POINTER(TransformF)]),
    "lio_est_get_prior": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int)]),
    "lio_est_get_prior_factor": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int)]),
    "lio_est_set_prior_factor": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int]),
    "lio_est_set_extrinsic": (C.c_int, [C.c_void_p, C.POINTER(TransformF)]),
    "lio_dense_spd_solve": (C.c_int, [c_double_p, c_double_p, C.c_int, c_double_p]),
    "lio_marginalize_schur": (C.c_int, [c_double_p, c_double_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    "lio_est_snapshot": (C.c_int, [C.c_void_p]),
    "lio_est_restore": (C.c_int, [C.c_void_p]),
    "lio_est_solve_restored": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SolveReport)]),
    "lio_pp_get_ring_intensity": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "lio_pp_check_config": (C.c_int, [C.c_float, C.c_float, C.c_int, C.POINTER(PPConfig)]),
    "lio_est_copy_snapshot": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lio_est_batch_create": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int]),
    "lio_est_batch_destroy": (None, [C.c_void_p]),
    "lio_est_batch_size": (C.c_int, [C.c_void_p]),
    "lio_est_batch_solve": (C.c_int, [C.c_void_p, C.POINTER(SolveReport)]),
    "lio_est_batch_solve_restored": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SolveReport)]),
    "lio_est_batch_sync": (C.c_int, [C.c_void_p]),
    "lio_est_batch_get_clock": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "lio_est_batch_stage_digest": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]),
    "lio_est_batch_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "lio_seg_sort_pairs": (C.c_int, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int,
                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "lio_est_set_factor_sharding": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "lio_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "lio_rccl_init": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int]),
    "lio_rccl_destroy": (None, [C.c_void_p]),
    "lio_rccl_rank": (C.c_int, [C.c_void_p]),
    "lio_rccl_world": (C.c_int, [C.c_void_p]),
    "lio_est_set_factor_sharding_rccl": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lio_kf_batch_refine_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_float_p, c_double_p]),
    "lio_rccl_bench_all_reduce": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p]),
    "lio_est_bench_batched_moments": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p]),
    "lio_est_enable_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "lio_est_get_kernel_timing": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p, c_double_p]),
}


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _fp(a):
    return None if a is None else a.ctypes.data_as(c_float_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class LioError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise LioError(f"{what} failed with code {rc}")


class LioLib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise LioError(f"{path} is missing — build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.dll, name)
            fn.restype = res
            fn.argtypes = args
        self.backend = self.dll.lio_backend().decode()

    # ---- ImuInitializer (host math)
    @staticmethod
    def _laser_frames(transforms, pims):
        n = len(transforms)
        T = (TransformF * n)(*[TransformF.make(q, p) for q, p in transforms])
        P = (C.c_void_p * n)(*[(p.h if p is not None else None) for p in pims])
        return n, T, P

    def imu_estimate_extrinsic_rotation(self, transforms, pims, T_lb):
        """ImuInitializer::EstimateExtrinsicRotation -> (accepted, q_lb xyzw)."""
        n, T, P = self._laser_frames(transforms, pims)
        lb = TransformF.make(*T_lb)
        rc = self.dll.lio_imu_estimate_extrinsic_rotation(n, T, P, C.byref(lb))
        if rc < 0:
            raise LioError(f"lio_imu_estimate_extrinsic_rotation -> {rc}")
        return bool(rc), lb.to_np()[0]

    def imu_initialization(self, transforms, pims, T_lb, Bgs=None):
        """ImuInitializer::Initialization -> dict(ok, Vs, Bgs, g, R_WI); re-propagates the pims."""
        n, T, P = self._laser_frames(transforms, pims)
        lb = TransformF.make(*T_lb)
        Vs = np.zeros((n, 3))
        bgs = np.zeros((n, 3)) if Bgs is None else _f64(Bgs).reshape(n, 3).copy()
        g, R = np.zeros(3), np.zeros((3, 3))
        rc = self.dll.lio_imu_initialization(n, T, P, C.byref(lb), _dp(Vs), _dp(bgs), _dp(g), _dp(R))
        if rc < 0:
            raise LioError(f"lio_imu_initialization -> {rc}")
        return dict(ok=bool(rc), Vs=Vs, Bgs=bgs, g=g, R_WI=R)

    # ---- stateless blocks
    def dense_spd_solve(self, A, b):
        A, b = _f64(A), _f64(b)
        n = b.shape[0]
        assert A.shape == (n, n)
        x = np.zeros(n)
        _chk(self.dll.lio_dense_spd_solve(_dp(A), _dp(b), n, _dp(x)), "lio_dense_spd_solve")
        return x

    def marginalize_schur(self, A, b, m):
        """-> (lin_jac n x n, lin_res n, evals n) of MarginalizationInfo::Marginalize's dense tail (MarginalizationFactor.cc:271-302)."""
        A, b = _f64(A), _f64(b)
        n = b.shape[0] - m
        assert A.shape == (m + n, m + n)
        J, r, s = np.zeros((n, n)), np.zeros(n), np.zeros(n)
        _chk(self.dll.lio_marginalize_schur(_dp(A), _dp(b), m, n, _dp(J), _dp(r), _dp(s)), "lio_marginalize_schur")
        return J, r, s

    def bench_voxel_grid(self, xyzi, leaf, reps=10):
        """device time of one VoxelGrid of a resident cloud (HIP events over `reps` runs) -> (ms, output points)"""
        xyzi = _f32(xyzi).reshape(-1, 4)
        ms, n_out = C.c_double(0), C.c_size_t(0)
        _chk(self.dll.lio_bench_voxel_grid(_fp(xyzi), xyzi.shape[0], float(leaf), int(reps), C.byref(ms), C.byref(n_out)), "lio_bench_voxel_grid")
        return ms.value, int(n_out.value)

    def seg_sort_pairs(self, keys, values, seg_off, seg_n, bits, passes):
        """stable radix sort of (key, value) pairs inside segments (lio_seg_sort_pairs); values None: the elements' positions"""
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        vals = None if values is None else np.ascontiguousarray(values, dtype=np.uint32)
        so, sn = np.ascontiguousarray(seg_off, dtype=np.int32), np.ascontiguousarray(seg_n, dtype=np.int32)
        ko, vo = np.zeros_like(keys), np.zeros_like(keys)
        u32 = C.POINTER(C.c_uint32)
        _chk(self.dll.lio_seg_sort_pairs(keys.ctypes.data_as(u32), vals.ctypes.data_as(u32) if vals is not None else None, keys.shape[0],
                                         so.ctypes.data_as(c_int32_p), sn.ctypes.data_as(c_int32_p), so.shape[0], int(bits), int(passes),
                                         ko.ctypes.data_as(u32), vo.ctypes.data_as(u32)), "lio_seg_sort_pairs")
        return ko, vo

    def voxel_grid(self, xyzi, leaf):
        xyzi = _f32(xyzi).reshape(-1, 4)
        out = np.zeros_like(xyzi)
        n = C.c_size_t(0)
        _chk(self.dll.lio_voxel_grid(_fp(xyzi), xyzi.shape[0], leaf, _fp(out), C.byref(n)), "lio_voxel_grid")
        return out[: n.value].copy()

    def knn(self, map_xyzi, query_xyzi, k, radius_sq=0.0):
        m_ = _f32(map_xyzi).reshape(-1, 4)
        q_ = _f32(query_xyzi).reshape(-1, 4)
        idx = np.zeros((q_.shape[0], k), dtype=np.int32)
        sqd = np.zeros((q_.shape[0], k), dtype=np.float32)
        _chk(
            self.dll.lio_knn(_fp(m_), m_.shape[0], _fp(q_), q_.shape[0], k, radius_sq, idx.ctypes.data_as(c_int32_p), _fp(sqd)), "lio_knn"
        )
        return idx, sqd

    def calculate_features(self, map_xyzi, stack_xyzi, T: TransformF, min_match_sq_dis=1.0, min_plane_dis=0.2):
        m_ = _f32(map_xyzi).reshape(-1, 4)
        s_ = _f32(stack_xyzi).reshape(-1, 4)
        valid = np.zeros(s_.shape[0], dtype=np.uint8)
        coeff = np.zeros((s_.shape[0], 4), dtype=np.float32)
        score = np.zeros(s_.shape[0], dtype=np.float32)
        _chk(
            self.dll.lio_calculate_features(
                _fp(m_), m_.shape[0], _fp(s_), s_.shape[0], C.byref(T), min_match_sq_dis, min_plane_dis, valid.ctypes.data_as(c_uint8_p), _fp(coeff), _fp(score)
            ),
            "lio_calculate_features",
        )
        return valid, coeff, score

    # ---- /compact_data codec
    def compact_encode(self, T: TransformF, corner, surf, full):
        cl = [_f32(c).reshape(-1, 4) for c in (corner, surf, full)]
        out = np.zeros((3 + sum(c.shape[0] for c in cl), 4), dtype=np.float32)
        args = []
        for c in cl:
            args += [_fp(c), c.shape[0]]
        n = self.dll.lio_compact_encode(C.byref(T), *args, _fp(out))
        assert n == out.shape[0]
        return out

    def compact_decode(self, data):
        data = _f32(data).reshape(-1, 4)
        T = TransformF()
        nc, ns, nf = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        rc = self.dll.lio_compact_decode(_fp(data), data.shape[0], C.byref(T), C.byref(nc), C.byref(ns), C.byref(nf))
        if rc != 0:
            raise LioError(f"lio_compact_decode failed with code {rc}")
        a, b = 3 + nc.value, 3 + nc.value + ns.value
        return T.to_np(), data[3:a], data[a:b], data[b:]

    # ---- factors
    def factor_ppp(self, point, coeff, pose_p, pose_i, pose_ex, jac=True):
        res = np.zeros(1)
        js = [np.zeros(7) for _ in range(3)] if jac else [None] * 3
        _chk(
            self.dll.lio_factor_pivot_point_plane(
                _dp(_f64(point)), _dp(_f64(coeff)), _dp(_f64(pose_p)), _dp(_f64(pose_i)), _dp(_f64(pose_ex)), _dp(res), *[_dp(j) for j in js]
            ),
            "lio_factor_pivot_point_plane",
        )
        return res[0], js

    def factor_prior(self, pos0, rot0, pose, jac=True):
        res = np.zeros(6)
        J = np.zeros((6, 7)) if jac else None
        _chk(self.dll.lio_factor_prior(_dp(_f64(pos0)), _dp(_f64(rot0)), _dp(_f64(pose)), _dp(res), _dp(J)), "lio_factor_prior")
        return res, J

    def pose_plus(self, pose, delta):
        out = np.zeros(7)
        _chk(self.dll.lio_pose_plus(_dp(_f64(pose)), _dp(_f64(delta)), _dp(out)), "lio_pose_plus")
        return out

    def default_est_config(self) -> EstConfig:
        c = EstConfig()
        self.dll.lio_est_default_config(C.byref(c))
        return c


class Pim:
    def __init__(self, lib: LioLib, acc0, gyr0, ba, bg, acc_n=0.1, gyr_n=0.01, acc_w=0.0002, gyr_w=2.0e-5, g_norm=9.805):
        self.lib = lib
        self.h = lib.dll.lio_pim_create(_dp(_f64(acc0)), _dp(_f64(gyr0)), _dp(_f64(ba)), _dp(_f64(bg)), acc_n, gyr_n, acc_w, gyr_w, g_norm)
        if not self.h:
            raise LioError("lio_pim_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_pim_destroy(self.h)
            self.h = None

    def push_back(self, dt, acc, gyr):
        _chk(self.lib.dll.lio_pim_push_back(self.h, dt, _dp(_f64(acc)), _dp(_f64(gyr))), "lio_pim_push_back")

    def repropagate(self, ba, bg):
        _chk(self.lib.dll.lio_pim_repropagate(self.h, _dp(_f64(ba)), _dp(_f64(bg))), "lio_pim_repropagate")

    def get(self):
        sum_dt = np.zeros(1)
        dp, dq, dv = np.zeros(3), np.zeros(4), np.zeros(3)
        jac, cov = np.zeros((15, 15)), np.zeros((15, 15))
        _chk(self.lib.dll.lio_pim_get(self.h, _dp(sum_dt), _dp(dp), _dp(dq), _dp(dv), _dp(jac), _dp(cov)), "lio_pim_get")
        return dict(sum_dt=sum_dt[0], dp=dp, dq=dq, dv=dv, jac=jac, cov=cov)

    def evaluate(self, pose_i, sb_i, pose_j, sb_j):
        res = np.zeros(15)
        _chk(self.lib.dll.lio_pim_evaluate(self.h, _dp(_f64(pose_i)), _dp(_f64(sb_i)), _dp(_f64(pose_j)), _dp(_f64(sb_j)), _dp(res)), "lio_pim_evaluate")
        return res

    def factor(self, pose_i, sb_i, pose_j, sb_j, jac=True):
        res = np.zeros(15)
        js = [np.zeros((15, 7)), np.zeros((15, 9)), np.zeros((15, 7)), np.zeros((15, 9))] if jac else [None] * 4
        _chk(
            self.lib.dll.lio_factor_imu(self.h, _dp(_f64(pose_i)), _dp(_f64(sb_i)), _dp(_f64(pose_j)), _dp(_f64(sb_j)), _dp(res), *[_dp(j) for j in js]),
            "lio_factor_imu",
        )
        return res, js


class PointProcessor:
    RINGS, SHARP, LESS_SHARP, FLAT, LESS_FLAT = range(5)

    def __init__(self, lib: LioLib, lower, upper, rings, cfg: PPConfig | None = None):
        self.lib = lib
        self.rings = rings
        self.h = lib.dll.lio_pp_create(lower, upper, rings, C.byref(cfg) if cfg is not None else None)
        if not self.h:
            raise LioError("lio_pp_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_pp_destroy(self.h)
            self.h = None

    def process_async(self, xyzi):
        """enqueue a sweep and return; `wait()` (or any accessor) blocks until it is done.  Keeps `xyzi` alive meanwhile."""
        self._pending = _f32(xyzi).reshape(-1, 4)
        _chk(self.lib.dll.lio_pp_process_async(self.h, _fp(self._pending), self._pending.shape[0]), "lio_pp_process_async")

    def wait(self):
        _chk(self.lib.dll.lio_pp_wait(self.h), "lio_pp_wait")
        self._pending = None

    @staticmethod
    def process_batch(processors, sweeps):
        """lio_pp_process_batch: one sweep per handle, all enqueued before the first is waited for."""
        assert len(processors) == len(sweeps)
        B = len(processors)
        if B == 0:
            return
        arrs = [_f32(x).reshape(-1, 4) for x in sweeps]
        hs = (C.c_void_p * B)(*[p.h for p in processors])
        ptrs = (c_float_p * B)(*[_fp(a) for a in arrs])
        ns = (C.c_size_t * B)(*[a.shape[0] for a in arrs])
        _chk(processors[0].lib.dll.lio_pp_process_batch(hs, ptrs, ns, B), "lio_pp_process_batch")

    @staticmethod
    def process_rings_batch(processors, sweeps, rings):
        """lio_pp_process_rings_batch: the PointIR overload (a ring per point) for B sweeps in one call."""
        B = len(processors)
        assert B == len(sweeps) == len(rings)
        if B == 0:
            return
        arrs = [_f32(x).reshape(-1, 4) for x in sweeps]
        rgs = [np.ascontiguousarray(r, dtype=np.uint16) for r in rings]
        hs = (C.c_void_p * B)(*[p.h for p in processors])
        ptrs = (c_float_p * B)(*[_fp(a) for a in arrs])
        rptrs = (c_uint16_p * B)(*[r.ctypes.data_as(c_uint16_p) for r in rgs])
        ns = (C.c_size_t * B)(*[a.shape[0] for a in arrs])
        _chk(processors[0].lib.dll.lio_pp_process_rings_batch(hs, ptrs, rptrs, ns, B), "lio_pp_process_rings_batch")

    @staticmethod
    def process_batch_device(processors, device_ptrs, counts):
        """lio_pp_process_batch_device: the sweeps are already in device memory (device_ptrs[k]: address of counts[k] x 4 floats)."""
        B = len(processors)
        assert B == len(device_ptrs) == len(counts)
        if B == 0:
            return
        hs = (C.c_void_p * B)(*[p.h for p in processors])
        ptrs = (C.c_void_p * B)(*[int(a) for a in device_ptrs])
        ns = (C.c_size_t * B)(*[int(c) for c in counts])
        _chk(processors[0].lib.dll.lio_pp_process_batch_device(hs, ptrs, ns, B), "lio_pp_process_batch_device")

    def process(self, xyzi, ring=None):
        """ring (uint16 per point) selects the PointIR overload of PointToRing (uneven sensors, PointProcessor.cc:428-536)."""
        xyzi = _f32(xyzi).reshape(-1, 4)
        if ring is None:
            _chk(self.lib.dll.lio_pp_process(self.h, _fp(xyzi), xyzi.shape[0]), "lio_pp_process")
            return
        ring = np.ascontiguousarray(ring, dtype=np.uint16)
        assert ring.shape[0] == xyzi.shape[0]
        _chk(self.lib.dll.lio_pp_process_rings(self.h, _fp(xyzi), ring.ctypes.data_as(c_uint16_p), xyzi.shape[0]), "lio_pp_process_rings")

    def start_ori(self):
        """start_ori_ of the last process call (after infer_start_ori's filter when that is enabled)."""
        return float(self.lib.dll.lio_pp_start_ori(self.h))

    def cloud(self, which):
        n = self.lib.dll.lio_pp_count(self.h, which)
        out = np.zeros((n, 4), dtype=np.float32)
        if n:
            _chk(self.lib.dll.lio_pp_get_cloud(self.h, which, _fp(out)), "lio_pp_get_cloud")
        return out

    def indices(self, which):
        n = self.lib.dll.lio_pp_count(self.h, which)
        ring = np.zeros(n, dtype=np.int32)
        idx = np.zeros(n, dtype=np.int32)
        if n:
            _chk(self.lib.dll.lio_pp_get_indices(self.h, which, ring.ctypes.data_as(c_int32_p), idx.ctypes.data_as(c_int32_p)), "lio_pp_get_indices")
        return ring, idx

    def ring_offsets(self):
        out = np.zeros(self.rings + 1, dtype=np.int32)
        _chk(self.lib.dll.lio_pp_get_ring_offsets(self.h, out.ctypes.data_as(c_int32_p)), "lio_pp_get_ring_offsets")
        return out

    def ring_intensity(self):
        """intensity channel of the reference's intensity_scans, ring order: int(input intensity) + rel_time"""
        out = np.zeros(int(self.lib.dll.lio_pp_count(self.h, self.RINGS)), np.float32)
        _chk(self.lib.dll.lio_pp_get_ring_intensity(self.h, _fp(out)), "lio_pp_get_ring_intensity")
        return out

    def curvature(self):
        n = self.lib.dll.lio_pp_count(self.h, 0)
        curv = np.zeros(n, dtype=np.float32)
        mask = np.zeros(n, dtype=np.int32)
        _chk(self.lib.dll.lio_pp_get_curvature(self.h, _fp(curv), mask.ctypes.data_as(c_int32_p)), "lio_pp_get_curvature")
        return curv, mask


class PointMapping:
    """PointMapping (scan-to-map + cube map) — reference src/point_processor/PointMapping.cc."""

    CORNER_STACK_DS, SURF_STACK_DS, CORNER_FROM_MAP, SURF_FROM_MAP = 0, 1, 2, 3

    def __init__(self, lib: LioLib, **overrides):
        self.lib = lib
        cfg = MapConfig()
        lib.dll.lio_map_default_config(C.byref(cfg))
        for k, v in overrides.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = lib.dll.lio_map_create(C.byref(cfg))
        if not self.h:
            raise LioError("lio_map_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_map_destroy(self.h)
            self.h = None

    def process(self, corner_last, surf_last, T_sum):
        c, s = _f32(corner_last).reshape(-1, 4), _f32(surf_last).reshape(-1, 4)
        Ts = TransformF.make(*T_sum)
        Ta = TransformF()
        it, ns = C.c_int(0), C.c_int(0)
        _chk(self.lib.dll.lio_map_process(self.h, _fp(c), c.shape[0], _fp(s), s.shape[0], C.byref(Ts), C.byref(Ta), C.byref(it), C.byref(ns)),
             "lio_map_process")
        kz = C.c_int(0)
        deg = self.lib.dll.lio_map_get_degeneracy(self.h, C.byref(kz))
        return dict(T_aft=Ta.to_np(), iterations=it.value, num_selected=ns.value, degenerate=int(deg), kz=kz.value)

    def set_init_flag(self, on):
        _chk(self.lib.dll.lio_map_set_init_flag(self.h, 1 if on else 0), "lio_map_set_init_flag")

    def set_transform_tobe_mapped(self, q_xyzw, p):
        T = TransformF.make(q_xyzw, p)
        _chk(self.lib.dll.lio_map_set_transform_tobe_mapped(self.h, C.byref(T)), "lio_map_set_transform_tobe_mapped")

    def transform_tobe_mapped(self):
        T = TransformF()
        _chk(self.lib.dll.lio_map_get_transform_tobe_mapped(self.h, C.byref(T)), "lio_map_get_transform_tobe_mapped")
        return T.to_np()

    def update_map_database(self, corner_ds, surf_ds, valid_idx, T, cube_center):
        c, s = _f32(corner_ds).reshape(-1, 4), _f32(surf_ds).reshape(-1, 4)
        v = np.ascontiguousarray(valid_idx, dtype=np.uint32)
        Tt = TransformF.make(*T)
        cen = (C.c_int * 3)(*[int(x) for x in cube_center])
        _chk(self.lib.dll.lio_map_update_map_database(self.h, _fp(c), c.shape[0], _fp(s), s.shape[0], v.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                      v.shape[0], C.byref(Tt), cen), "lio_map_update_map_database")

    def cloud(self, which):
        n = self.lib.dll.lio_map_get_cloud(self.h, which, None)
        out = np.zeros((n, 4), dtype=np.float32)
        if n:
            self.lib.dll.lio_map_get_cloud(self.h, which, _fp(out))
        return out

    def cube(self, cls, cube_idx):
        n = self.lib.dll.lio_map_get_cube(self.h, cls, int(cube_idx), None)
        out = np.zeros((n, 4), dtype=np.float32)
        if n:
            self.lib.dll.lio_map_get_cube(self.h, cls, int(cube_idx), _fp(out))
        return out

    def cube_state(self):
        cen = (C.c_int * 3)()
        valid = np.zeros(125, dtype=np.uint32)
        n = self.lib.dll.lio_map_get_cube_state(self.h, cen, valid.ctypes.data_as(C.POINTER(C.c_uint32)))
        return list(cen), valid[:n].copy()

    def score_point_coeff(self):
        n = self.lib.dll.lio_map_get_score_point_coeff(self.h, None, None, None)
        score = np.zeros(n, dtype=np.float32)
        point = np.zeros((n, 4), dtype=np.float32)
        coeff = np.zeros((n, 4), dtype=np.float32)
        if n:
            self.lib.dll.lio_map_get_score_point_coeff(self.h, _fp(score), _fp(point), _fp(coeff))
        return score, point, coeff


class Rccl:
    """In-library RCCL communicator (one per process / GPU): `Rccl.unique_id(lib)` on rank 0, the bytes sent to every rank by
    any side channel, then `Rccl(lib, id_bytes, rank, world)` on the device the process drives."""

    @staticmethod
    def unique_id(lib: LioLib) -> bytes:
        buf = C.create_string_buffer(128)
        _chk(lib.dll.lio_rccl_unique_id(buf), "lio_rccl_unique_id")
        return bytes(buf.raw)

    def __init__(self, lib: LioLib, id_bytes: bytes, rank: int, world: int):
        assert len(id_bytes) == 128
        self.lib = lib
        self.h = lib.dll.lio_rccl_init(C.create_string_buffer(id_bytes, 128), int(rank), int(world))
        if not self.h:
            raise LioError("lio_rccl_init failed")
        self.rank, self.world = int(rank), int(world)

    def world_seen_by_rccl(self) -> int:
        """lio_rccl_world: what the communicator itself reports (bench.py prints it next to n_gpus)."""
        return int(self.lib.dll.lio_rccl_world(self.h))

    def bench_all_reduce(self, count: int, reps: int = 200) -> float:
        """collective: mean microseconds per in-place SUM all-reduce of `count` doubles (HIP events, this rank)."""
        us = C.c_double(0)
        _chk(self.lib.dll.lio_rccl_bench_all_reduce(self.h, int(count), int(reps), C.byref(us)), "lio_rccl_bench_all_reduce")
        return us.value

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_rccl_destroy(self.h)
            self.h = None


class KeyframeBatch:
    """Batched keyframe refinement (BASELINE.json configs[4]): one OptimizeMap / OptimizeTransformTobeMapped loop per
    keyframe (MapBuilder.cc:624-1014 / PointMapping.cc:325-753), all keyframes of the handle advanced together."""

    def __init__(self, lib: LioLib, **overrides):
        self.lib = lib
        cfg = MapConfig()
        lib.dll.lio_map_default_config(C.byref(cfg))
        for k, v in overrides.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = lib.dll.lio_kf_batch_create(C.byref(cfg))
        if not self.h:
            raise LioError("lio_kf_batch_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_kf_batch_destroy(self.h)
            self.h = None

    def add_map(self, corner_map, surf_map):
        c, s = _f32(corner_map).reshape(-1, 4), _f32(surf_map).reshape(-1, 4)
        idx = self.lib.dll.lio_kf_batch_add_map(self.h, _fp(c), c.shape[0], _fp(s), s.shape[0])
        if idx < 0:
            _chk(idx, "lio_kf_batch_add_map")
        return idx

    def add_keyframe(self, map_index, corner_stack, surf_stack, T_init):
        c, s = _f32(corner_stack).reshape(-1, 4), _f32(surf_stack).reshape(-1, 4)
        T = TransformF.make(*T_init)
        idx = self.lib.dll.lio_kf_batch_add_keyframe(self.h, int(map_index), _fp(c), c.shape[0], _fp(s), s.shape[0], C.byref(T))
        if idx < 0:
            _chk(idx, "lio_kf_batch_add_keyframe")
        return idx

    def clear_keyframes(self):
        _chk(self.lib.dll.lio_kf_batch_clear_keyframes(self.h), "lio_kf_batch_clear_keyframes")

    def __len__(self):
        return int(self.lib.dll.lio_kf_batch_size(self.h))

    def refine_gather(self, comm: "Rccl", slots_per_rank: int):
        """refine() + all-gather of every rank's results inside the library (ncclAllGather from the device pose buffer).
        -> (world, slots_per_rank, 9) float32: q xyzw, p, iterations, rows; rows beyond a rank's keyframe count are zero."""
        out = np.zeros((comm.world, int(slots_per_rank), 9), dtype=np.float32)
        ms = C.c_double(0)
        _chk(self.lib.dll.lio_kf_batch_refine_gather(self.h, comm.h, int(slots_per_rank), _fp(out), C.byref(ms)), "lio_kf_batch_refine_gather")
        self.last_device_ms = ms.value
        return out

    def refine(self):
        """-> dict(q (n,4) xyzw, p (n,3), iterations (n,), rows (n,), device_ms)"""
        n = len(self)
        T = (TransformF * max(n, 1))()
        it = np.zeros(max(n, 1), dtype=np.int32)
        rows = np.zeros(max(n, 1), dtype=np.int32)
        ms = C.c_double(0)
        _chk(self.lib.dll.lio_kf_batch_refine(self.h, T, it.ctypes.data_as(c_int32_p), rows.ctypes.data_as(c_int32_p), C.byref(ms)),
             "lio_kf_batch_refine")
        arr = np.frombuffer(T, dtype=np.float32).reshape(-1, 7)[:n]
        kz = np.zeros(max(n, 1), dtype=np.int32)
        _chk(self.lib.dll.lio_kf_batch_get_degeneracy(self.h, kz.ctypes.data_as(c_int32_p)), "lio_kf_batch_get_degeneracy")
        return dict(q=arr[:, :4].copy(), p=arr[:, 4:7].copy(), iterations=it[:n], rows=rows[:n], kz=kz[:n], device_ms=ms.value)


class PointOdometry:
    def __init__(self, lib: LioLib, scan_period=0.1, io_ratio=2, max_iter=25, no_deskew=False):
        self.lib = lib
        self.h = lib.dll.lio_odom_create(scan_period, io_ratio, max_iter, 1 if no_deskew else 0)
        if not self.h:
            raise LioError("lio_odom_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_odom_destroy(self.h)
            self.h = None

    def process(self, sharp, less_sharp, flat, less_flat):
        cl = [_f32(c).reshape(-1, 4) for c in (sharp, less_sharp, flat, less_flat)]
        Ts, Te = TransformF(), TransformF()
        it, ns = C.c_int(0), C.c_int(0)
        args = []
        for c in cl:
            args += [_fp(c), c.shape[0]]
        _chk(self.lib.dll.lio_odom_process(self.h, *args, C.byref(Ts), C.byref(Te), C.byref(it), C.byref(ns)), "lio_odom_process")
        cap = max(it.value, 1)
        tr = (TransformF * cap)()
        kz = C.c_int(0)
        n = self.lib.dll.lio_odom_get_iteration_trace(self.h, tr, cap, C.byref(kz))
        if n < 0:
            raise LioError(f"lio_odom_get_iteration_trace -> {n}")
        trace = np.frombuffer(tr, dtype=np.float32).reshape(-1, 7)[:n].copy()   # rows: q xyzw, p
        return dict(T_sum=Ts.to_np(), T_es=Te.to_np(), iterations=it.value, num_selected=ns.value, trace=trace, kz=kz.value)

    def enable(self, on):
        _chk(self.lib.dll.lio_odom_enable(self.h, 1 if on else 0), "lio_odom_enable")

    def last_cloud(self, which):
        n = self.lib.dll.lio_odom_get_last_cloud(self.h, which, None)
        out = np.zeros((n, 4), dtype=np.float32)
        if n:
            self.lib.dll.lio_odom_get_last_cloud(self.h, which, _fp(out))
        return out


class Estimator:
    def __init__(self, lib: LioLib, cfg: EstConfig):
        self.lib = lib
        self.cfg = cfg
        self.W = cfg.window_size
        self.h = lib.dll.lio_est_create(C.byref(cfg))
        if not self.h:
            raise LioError("lio_est_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_est_destroy(self.h)
            self.h = None

    def process_imu(self, dt, acc, gyr, stamp):
        _chk(self.lib.dll.lio_est_process_imu(self.h, dt, _dp(_f64(acc)), _dp(_f64(gyr)), stamp), "lio_est_process_imu")

    def process_imu_batch(self, dt, acc, gyr, stamp):
        """All IMU samples of one laser interval in one call (the loop of Estimator::ProcessEstimation)."""
        dt, stamp = _f64(dt).reshape(-1), _f64(stamp).reshape(-1)
        acc, gyr = _f64(acc).reshape(-1, 3), _f64(gyr).reshape(-1, 3)
        assert acc.shape[0] == gyr.shape[0] == dt.shape[0] == stamp.shape[0]
        _chk(self.lib.dll.lio_est_process_imu_batch(self.h, dt.shape[0], _dp(dt), _dp(acc), _dp(gyr), _dp(stamp)), "lio_est_process_imu_batch")

    def process_laser_odom(self, T: TransformF, surf, corner, stamp):
        surf = _f32(surf).reshape(-1, 4)
        corner = _f32(corner).reshape(-1, 4)
        rep = SolveReport()
        _chk(
            self.lib.dll.lio_est_process_laser_odom(self.h, C.byref(T), _fp(surf), surf.shape[0], _fp(corner), corner.shape[0], stamp, C.byref(rep)),
            "lio_est_process_laser_odom",
        )
        return rep

    def process_compact(self, compact, stamp):
        """Estimator::ProcessCompactData -> (transform_to_init (q, p), SolveReport)."""
        data = _f32(compact).reshape(-1, 4)
        rep = SolveReport()
        T = TransformF()
        _chk(self.lib.dll.lio_est_process_compact(self.h, _fp(data), data.shape[0], stamp, C.byref(T), C.byref(rep)), "lio_est_process_compact")
        return T.to_np(), rep

    EVENTS = ("skipped", "filling", "init_failed", "initialised", "solved")

    def stage(self):
        st, cb, ex, ev = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        R, g = np.zeros((3, 3)), np.zeros(3)
        _chk(self.lib.dll.lio_est_get_stage(self.h, C.byref(st), C.byref(cb), C.byref(ex), C.byref(ev), _dp(R), _dp(g)), "lio_est_get_stage")
        return dict(inited=bool(st.value), cir_buf_count=cb.value, extrinsic_stage=ex.value, event=self.EVENTS[ev.value], R_WI=R, g_vec=g)

    def push_frame(self, T: TransformF, surf, corner, stamp):
        surf = _f32(surf).reshape(-1, 4)
        corner = _f32(corner).reshape(-1, 4)
        _chk(self.lib.dll.lio_est_push_frame(self.h, C.byref(T), _fp(surf), surf.shape[0], _fp(corner), corner.shape[0], stamp), "lio_est_push_frame")

    def solve(self):
        rep = SolveReport()
        _chk(self.lib.dll.lio_est_solve_optimization(self.h, C.byref(rep)), "lio_est_solve_optimization")
        return rep

    def slide(self):
        _chk(self.lib.dll.lio_est_slide_window(self.h), "lio_est_slide_window")

    def sync(self):
        """Wait for the handle's deferred work (the marginalization worker of the product)."""
        _chk(self.lib.dll.lio_est_sync(self.h), "lio_est_sync")

    def set_window(self, Ps, Rs, Vs, Bas, Bgs, g_vec):
        n = self.W + 1
        Ps, Rs, Vs, Bas, Bgs = (_f64(a) for a in (Ps, Rs, Vs, Bas, Bgs))
        assert Ps.shape == (n, 3) and Rs.shape == (n, 3, 3)
        _chk(self.lib.dll.lio_est_set_window(self.h, n, _dp(Ps), _dp(Rs), _dp(Vs), _dp(Bas), _dp(Bgs), _dp(_f64(g_vec))), "lio_est_set_window")

    def get_window(self):
        n = self.W + 1
        Ps, Rs, Vs, Bas, Bgs = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        T = TransformF()
        _chk(self.lib.dll.lio_est_get_window(self.h, n, _dp(Ps), _dp(Rs), _dp(Vs), _dp(Bas), _dp(Bgs), C.byref(T)), "lio_est_get_window")
        q, p = T.to_np()
        return dict(Ps=Ps, Rs=Rs, Vs=Vs, Bas=Bas, Bgs=Bgs, q_lb=q, t_lb=p)

    def set_surf_stack(self, frame, xyzi):
        xyzi = _f32(xyzi).reshape(-1, 4)
        _chk(self.lib.dll.lio_est_set_surf_stack(self.h, frame, _fp(xyzi), xyzi.shape[0]), "lio_est_set_surf_stack")

    def get_surf_stack(self, frame):
        n = self.lib.dll.lio_est_get_surf_stack(self.h, frame, None)
        out = np.zeros((n, 4), dtype=np.float32)
        if n:
            self.lib.dll.lio_est_get_surf_stack(self.h, frame, _fp(out))
        return out

    def set_preintegration(self, frame, acc0, gyr0, ba, bg, dt, acc, gyr):
        dt, acc, gyr = _f64(dt), _f64(acc), _f64(gyr)
        _chk(
            self.lib.dll.lio_est_set_preintegration(
                self.h, frame, _dp(_f64(acc0)), _dp(_f64(gyr0)), _dp(_f64(ba)), _dp(_f64(bg)), _dp(dt), _dp(acc), _dp(gyr), dt.shape[0]
            ),
            "lio_est_set_preintegration",
        )

    def begin_frame(self, acc_last, gyr_last):
        _chk(self.lib.dll.lio_est_begin_frame(self.h, _dp(_f64(acc_last)), _dp(_f64(gyr_last))), "lio_est_begin_frame")

    def build_local_map(self):
        _chk(self.lib.dll.lio_est_build_local_map(self.h), "lio_est_build_local_map")

    def local_map(self):
        n = self.lib.dll.lio_est_get_local_map(self.h, None)
        out = np.zeros((n, 4), dtype=np.float32)
        if n:
            self.lib.dll.lio_est_get_local_map(self.h, _fp(out))
        return out

    def features(self, frame):
        n = self.lib.dll.lio_est_get_features(self.h, frame, None, None, None)
        pt, co, sc = np.zeros((n, 3)), np.zeros((n, 4)), np.zeros(n)
        if n:
            self.lib.dll.lio_est_get_features(self.h, frame, _dp(pt), _dp(co), _dp(sc))
        return pt, co, sc

    def laser_odom_transform(self):
        T = TransformF()
        _chk(self.lib.dll.lio_est_get_laser_odom_transform(self.h, C.byref(T)), "lio_est_get_laser_odom_transform")
        return T.to_np()

    def prior(self):
        n = self.lib.dll.lio_est_get_prior(self.h, None, None, None, None)
        if n <= 0:
            return None
        JtJ, Jtr = np.zeros((n, n)), np.zeros(n)
        ln = C.c_int(0)
        self.lib.dll.lio_est_get_prior(self.h, None, None, None, C.byref(ln))
        x0 = np.zeros(ln.value)
        self.lib.dll.lio_est_get_prior(self.h, _dp(JtJ), _dp(Jtr), _dp(x0), C.byref(ln))
        return dict(n=n, JtJ=JtJ, Jtr=Jtr, x0=x0)

    def prior_factor(self):
        """linearized_jacobians / linearized_residuals / keep_block_data of the current prior (None when there is none)."""
        n = self.lib.dll.lio_est_get_prior_factor(self.h, None, None, None, None)
        if n <= 0:
            return None
        ln = C.c_int(0)
        self.lib.dll.lio_est_get_prior_factor(self.h, None, None, None, C.byref(ln))
        J, r, x0 = np.zeros((n, n)), np.zeros(n), np.zeros(ln.value)
        self.lib.dll.lio_est_get_prior_factor(self.h, _dp(J), _dp(r), _dp(x0), C.byref(ln))
        return dict(n=n, lin_jac=J, lin_res=r, x0=x0)

    def set_prior_factor(self, pf):
        J, r, x0 = _f64(pf["lin_jac"]), _f64(pf["lin_res"]), _f64(pf["x0"])
        _chk(self.lib.dll.lio_est_set_prior_factor(self.h, int(pf["n"]), _dp(J), _dp(r), _dp(x0), x0.shape[0]), "lio_est_set_prior_factor")

    def set_extrinsic(self, q_xyzw, p):
        T = TransformF.make(q_xyzw, p)
        _chk(self.lib.dll.lio_est_set_extrinsic(self.h, C.byref(T)), "lio_est_set_extrinsic")

    def set_factor_sharding_rccl(self, comm):
        """comm: capi.Rccl or None.  The all-reduce of the per-shard moments then runs inside the library (ncclAllReduce on the
        estimator's stream)."""
        self._rccl = comm   # keep the communicator alive as long as the estimator uses it
        _chk(self.lib.dll.lio_est_set_factor_sharding_rccl(self.h, comm.h if comm is not None else None), "lio_est_set_factor_sharding_rccl")

    def set_factor_sharding(self, rank, world, allreduce_numpy):
        """`allreduce_numpy(buf: np.ndarray[float64])` must sum `buf` in place over all ranks (e.g. torch.distributed)."""
        ALLREDUCE = C.CFUNCTYPE(C.c_int, c_double_p, C.c_int, C.c_void_p)

        def _cb(ptr, count, _user):
            try:
                buf = np.ctypeslib.as_array(ptr, shape=(count,))
                allreduce_numpy(buf)
                return 0
            except Exception:  # noqa: BLE001 - must not propagate through the C frame
                return 1

        self._allreduce_cb = ALLREDUCE(_cb) if (world > 1 and allreduce_numpy is not None) else None  # keep alive
        fn = C.cast(self._allreduce_cb, C.c_void_p) if self._allreduce_cb else None
        _chk(self.lib.dll.lio_est_set_factor_sharding(self.h, rank, world, fn, None), "lio_est_set_factor_sharding")

    def bench_batched_moments(self, n_windows, reps=20):
        """One moments launch over n_windows copies of the current window's lidar factors -> (avg ms, algorithmic bytes)."""
        ms, b = np.zeros(1), np.zeros(1)
        _chk(self.lib.dll.lio_est_bench_batched_moments(self.h, int(n_windows), int(reps), _dp(ms), _dp(b)), "lio_est_bench_batched_moments")
        return float(ms[0]), float(b[0])

    def enable_kernel_timing(self, on=True):
        """on: False/0 stop, True/1 every launch, N > 1 every N-th launch of each kernel kind."""
        _chk(self.lib.dll.lio_est_enable_kernel_timing(self.h, int(on)), "lio_est_enable_kernel_timing")

    def kernel_timing(self, name):
        t, b = np.zeros(1), np.zeros(1)
        n = self.lib.dll.lio_est_get_kernel_timing(self.h, name.encode(), _dp(t), _dp(b))
        return dict(launches=int(n), total_ms=float(t[0]), algorithmic_bytes=float(b[0]))

    def snapshot(self):
        _chk(self.lib.dll.lio_est_snapshot(self.h), "lio_est_snapshot")

    def restore(self):
        _chk(self.lib.dll.lio_est_restore(self.h), "lio_est_restore")

    def copy_snapshot_of(self, src):
        """this handle's snapshot <- a copy of src's (lio_est_copy_snapshot)"""
        _chk(self.lib.dll.lio_est_copy_snapshot(self.h, src.h), "lio_est_copy_snapshot")

    def solve_restored(self, steps):
        """`steps` x (restore + SolveOptimization) inside the library -> the last solve's report"""
        rep = SolveReport()
        _chk(self.lib.dll.lio_est_solve_restored(self.h, int(steps), C.byref(rep)), "lio_est_solve_restored")
        return rep


class EstimatorBatch:
    """lio_est_batch: B estimators solved together, every stage one launch over all windows (include/lio_c.h).  The batch adopts
    the estimators; they must outlive it (this object keeps them referenced)."""

    def __init__(self, lib: LioLib, estimators):
        self.lib = lib
        self.members = list(estimators)
        arr = (C.c_void_p * len(self.members))(*[e.h for e in self.members])
        self.h = lib.dll.lio_est_batch_create(arr, len(self.members))
        if not self.h:
            raise LioError("lio_est_batch_create failed (null / duplicate / already adopted window, or no device)")

    def close(self):
        if getattr(self, "h", None):
            self.lib.dll.lio_est_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __len__(self):
        return int(self.lib.dll.lio_est_batch_size(self.h))

    def solve(self):
        reps = (SolveReport * len(self.members))()
        _chk(self.lib.dll.lio_est_batch_solve(self.h, reps), "lio_est_batch_solve")
        return list(reps)

    def solve_restored(self, steps):
        reps = (SolveReport * len(self.members))()
        _chk(self.lib.dll.lio_est_batch_solve_restored(self.h, int(steps), reps), "lio_est_batch_solve_restored")
        return list(reps)

    def sync(self):
        _chk(self.lib.dll.lio_est_batch_sync(self.h), "lio_est_batch_sync")

    STAGES = ("filtered_map", "knn_grid", "feature_flags", "plane_coefficients", "newest_frame_state", "solver_state", "moments", "jacobi_scaling", "scaled_hessian", "new_prior")

    def set_option(self, name, value):
        """an execution choice of the batch (lio_est_batch_set_option): results do not depend on it"""
        _chk(self.lib.dll.lio_est_batch_set_option(self.h, name.encode(), int(value)), f"lio_est_batch_set_option({name}, {value})")

    def stage_digest(self, stage):
        """one 64-bit digest per window of what stage `stage` (index into STAGES) of the last solve left on the device"""
        out = (C.c_ulonglong * len(self.members))()
        _chk(self.lib.dll.lio_est_batch_stage_digest(self.h, int(stage), out), "lio_est_batch_stage_digest")
        return np.array(list(out), dtype=np.uint64)

    def clock(self):
        out = (C.c_double * 24)()
        _chk(self.lib.dll.lio_est_batch_get_clock(self.h, out), "lio_est_batch_get_clock")
        names = ["describe", "filter", "grid_features_rounds", "pack", "solve", "finish", "fallback", "total", "n_device", "rounds",
                 "dev_filter", "dev_grid", "dev_features", "dev_rounds", "dev_loop", "dev_marg", "dev_marg_wait", "aux_ms", "moments_ms", "step_ms", "aux_launches", "moments_launches", "step_launches"]
        return dict(zip(names, [float(v) for v in out]))


def load_hip() -> LioLib:
    """The product library.  Fails loudly when it has not been built; never substitutes the oracle."""
    return LioLib(HIP_LIB_PATH)
