"""The C-ABI shared libraries load and export every symbol include/lio_c.h declares (no GPU needed), and
the product's host-side entry points agree with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch  # noqa: F401  -- before the product library is loaded: when both HIP users live in one process (only in this
#                              file: it asks torch whether a GPU exists), the runtime libraries must be loaded torch-first;
#                              the other order leaves the second one without a visible device (observed on the MI355X box)

from lio_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lio_c.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lio_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_in_python():
    syms = _declared_symbols()
    assert len(syms) > 40
    assert set(syms) == set(capi._SIGS.keys())


def test_hip_library_exports_every_symbol():
    assert os.path.exists(capi.HIP_LIB_PATH), "build the product first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = ctypes.CDLL(capi.HIP_LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(dll, s), s
    dll.lio_backend.restype = ctypes.c_char_p
    assert dll.lio_backend() == b"hip-gfx950"


def test_oracle_library_exports_every_symbol(oracle):
    for s in _declared_symbols():
        assert hasattr(oracle.dll, s), s
    assert oracle.backend == "oracle-cpu"


def _c_translation_unit(tmp_path, lib_path, backend):
    """include/lio_c.h compiled by a C compiler (not C++, not ctypes) and linked with the library."""
    import subprocess

    inc = tmp_path / "lio_symbols.inc"
    inc.write_text("".join(f"LIO_SYM({s})\n" for s in _declared_symbols()))
    exe = tmp_path / "c_abi_check"
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)
    assert libname.startswith("lib") and libname.endswith(".so")
    cmd = ["gcc", "-std=c99", "-pedantic-errors", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
           os.path.join(ROOT, "tests", "host", "c_abi_check.c"), "-o", str(exe), "-L", libdir, "-l" + libname[3:-3],
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    r = subprocess.run([str(exe), backend], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert backend in r.stdout


def test_header_is_plain_c_and_links_against_the_product(tmp_path):
    assert os.path.exists(capi.HIP_LIB_PATH)
    _c_translation_unit(tmp_path, capi.HIP_LIB_PATH, "hip-gfx950")


def test_header_is_plain_c_and_links_against_the_oracle(tmp_path, oracle):
    _c_translation_unit(tmp_path, os.path.join(ROOT, "oracle", "liblio_oracle.so"), "oracle-cpu")


def test_product_has_no_cpu_fallback(hip):
    """Without a GPU every data-path entry point must fail loudly (LIO_ERR_DEVICE), never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        return
    cfg = hip.default_est_config()
    assert not hip.dll.lio_est_create(ctypes.byref(cfg))
    out = np.zeros((4, 4), np.float32)
    n = ctypes.c_size_t(0)
    rc = hip.dll.lio_voxel_grid(out.ctypes.data_as(capi.c_float_p), 4, 0.4, out.ctypes.data_as(capi.c_float_p), ctypes.byref(n))
    assert rc == -3
    # every handle that owns device state refuses to exist
    assert not hip.dll.lio_map_create(None)
    assert not hip.dll.lio_kf_batch_create(None)
    assert not hip.dll.lio_odom_create(0.1, 2, 25, 0)
    assert not hip.dll.lio_pp_create(-15.0, 15.0, 16, None)


def test_host_side_preintegration_matches_oracle(hip, oracle):
    traj = synth.Trajectory()

    def mk(lib):
        p = capi.Pim(lib, traj.accel(1.0), traj.gyro(1.0), np.array([0.01, -0.02, 0.03]), np.array([0.001, 0.002, -0.001]), acc_n=0.2, gyr_n=0.02)
        for k in range(60):
            t = 1.0 + (k + 1) * 0.005
            p.push_back(0.005, traj.accel(t), traj.gyro(t))
        return p

    po, ph = mk(oracle), mk(hip)
    go, gh = po.get(), ph.get()
    for k in go:
        np.testing.assert_allclose(gh[k], go[k], rtol=1e-12, atol=1e-14)
    pose = lambda t: np.concatenate([traj.pos(t), synth.quat_from_rot(traj.rot(t))])
    rng = np.random.default_rng(0)
    pi, pj = oracle.pose_plus(pose(1.0), rng.normal(size=6) * 0.01), oracle.pose_plus(pose(1.3), rng.normal(size=6) * 0.01)
    sbi, sbj = np.concatenate([traj.vel(1.0), rng.normal(size=6) * 0.01]), np.concatenate([traj.vel(1.3), rng.normal(size=6) * 0.01])
    (ro, jo), (rh, jh) = po.factor(pi, sbi, pj, sbj), ph.factor(pi, sbi, pj, sbj)
    np.testing.assert_allclose(rh, ro, rtol=1e-10, atol=1e-9)
    for a, b in zip(jh, jo):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-8)
    po.repropagate(np.zeros(3), np.zeros(3))
    ph.repropagate(np.zeros(3), np.zeros(3))
    np.testing.assert_allclose(ph.get()["dp"], po.get()["dp"], rtol=1e-12)


def test_host_side_factors_match_oracle(hip, oracle):
    rng = np.random.default_rng(4)
    for _ in range(10):
        q = lambda: (lambda v: v / np.linalg.norm(v))(rng.normal(size=4))
        pp, pi = np.concatenate([rng.normal(size=3) * 3, q()]), np.concatenate([rng.normal(size=3) * 3, q()])
        pex = np.concatenate([rng.normal(size=3) * 0.3, q()])
        pt, co = rng.normal(size=3) * 10, rng.normal(size=4)
        (r1, j1), (r2, j2) = oracle.factor_ppp(pt, co, pp, pi, pex), hip.factor_ppp(pt, co, pp, pi, pex)
        np.testing.assert_allclose(r2, r1, rtol=1e-12, atol=1e-12)
        for a, b in zip(j2, j1):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
        (r1, J1), (r2, J2) = oracle.factor_prior(pex[:3], pex[3:], pi), hip.factor_prior(pex[:3], pex[3:], pi)
        np.testing.assert_allclose(r2, r1, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(J2, J1, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(hip.pose_plus(pp, co.repeat(2)[:6] * 0.01), oracle.pose_plus(pp, co.repeat(2)[:6] * 0.01), rtol=1e-14)


def test_compact_data_codec_round_trip(hip, oracle):
    """/compact_data (PointOdometry.cc:732-764 / PointMapping.cc:171-238): encode -> decode is the identity, both
    implementations produce the same bytes, malformed headers are rejected like the reference's LOG(ERROR) paths."""
    rng = np.random.default_rng(9)
    T = capi.TransformF.make([0.1, -0.2, 0.3, 0.92], [1.5, -2.5, 0.25])
    corner, surf, full = (rng.normal(size=(n, 4)).astype(np.float32) for n in (7, 0, 19))
    a, b = hip.compact_encode(T, corner, surf, full), oracle.compact_encode(T, corner, surf, full)
    np.testing.assert_array_equal(a, b)
    assert a.shape[0] == 3 + 7 + 0 + 19 and a[2, 3] == np.float32(0.92) and a[0, 3] == 0
    for lib in (hip, oracle):
        (q, p), c2, s2, f2 = lib.compact_decode(a)
        np.testing.assert_array_equal(c2, corner)
        assert s2.shape[0] == 0
        np.testing.assert_array_equal(f2, full)
        np.testing.assert_allclose(p, [1.5, -2.5, 0.25])
        bad = a.copy()
        bad[2, 0] += 1  # header no longer matches the payload
        try:
            lib.compact_decode(bad)
            assert False
        except capi.LioError:
            pass
        try:
            lib.compact_decode(a[:3])
            assert False
        except capi.LioError:
            pass
        # the sizes are wire floats: non-finite, negative, or absurdly large values are refused before any integer cast
        for col, val in ((0, np.nan), (1, np.inf), (2, -1.0), (0, 3.0e38), (1, 1.0e19)):
            bad = a.copy()
            bad[2, col] = val
            with pytest.raises(capi.LioError):
                lib.compact_decode(bad)


def test_error_codes_of_the_init_and_mapping_entry_points(hip, oracle):
    """Argument validation happens before any device work, so it is identical in both libraries: LIO_ERR_ARG (-1) for
    null / short / inconsistent inputs, NULL handles for bad configurations, never an exception across the C boundary."""
    import torch

    T = (capi.TransformF * 3)()
    pims = (ctypes.c_void_p * 3)(None, None, None)
    lb = capi.TransformF.make([0, 0, 0, 1], [0, 0, 0])
    Vs, Bgs, g, R = np.zeros(9), np.zeros(9), np.zeros(3), np.zeros(9)
    dp = lambda a: a.ctypes.data_as(capi.c_double_p)
    for lib in (oracle, hip):
        d = lib.dll
        assert d.lio_imu_estimate_extrinsic_rotation(1, T, pims, ctypes.byref(lb)) == -1            # n < 2
        assert d.lio_imu_estimate_extrinsic_rotation(3, T, pims, ctypes.byref(lb)) == -1            # pims[1] missing
        assert d.lio_imu_initialization(3, T, pims, ctypes.byref(lb), dp(Vs), dp(Bgs), dp(g), dp(R)) == -1
        assert d.lio_imu_initialization(3, T, None, ctypes.byref(lb), dp(Vs), dp(Bgs), dp(g), dp(R)) == -1
        bad = capi.MapConfig()
        d.lio_map_default_config(ctypes.byref(bad))
        assert (bad.map_builder, bad.enable_4d, bad.skip_count, bad.num_max_iterations) == (0, 1, 2, 10)
        bad.surf_filter_size = 0.0
        assert not d.lio_map_create(ctypes.byref(bad))
        bad.surf_filter_size, bad.map_builder, bad.skip_count = 0.4, 1, 0
        assert not d.lio_map_create(ctypes.byref(bad))
        assert d.lio_map_process(None, None, 0, None, 0, None, None, None, None) == -1
        assert d.lio_map_get_cloud(None, 0, None) == 0 and d.lio_map_get_cube_state(None, None, None) == 0
        assert d.lio_est_process_compact(None, None, 0, 0.0, None, None) == -1
        assert d.lio_est_get_stage(None, None, None, None, None, None, None) == -1
        cfg = lib.default_est_config()
        assert cfg.init_window_factor == 3 and cfg.extrinsic_stage == 2                              # Estimator.h:80-81
    # a malformed /compact_data message is rejected before anything is pushed (oracle always; product when a GPU exists)
    libs = [oracle] + ([hip] if torch.cuda.is_available() else [])
    for lib in libs:
        est = capi.Estimator(lib, lib.default_est_config())
        junk = np.zeros((5, 4), np.float32)
        junk[2, :3] = (3, 3, 3)        # claims 9 points, carries 2
        rc = lib.dll.lio_est_process_compact(est.h, junk.ctypes.data_as(capi.c_float_p), 5, 1.0, None, None)
        assert rc == -1
        assert est.stage()["cir_buf_count"] == 0 and est.stage()["event"] == "skipped"


def test_point_processor_capacity_is_reported_as_such(hip, oracle):
    """k_ring_pick keeps a subregion's picks one per lane of a wave: quotas above 64 picks are LIO_ERR_CAPACITY (-4), not a silent
    NULL; bad values are LIO_ERR_ARG; the reference's presets and defaults are accepted.  The oracle has no such limit."""
    cfg = capi.PPConfig()
    hip.dll.lio_pp_default_config(cfg)
    for lo, up, rings in ((-15.0, 15.0, 16), (-30.67, 10.67, 32), (-24.9, 2.0, 64), (-25.0, 15.0, 32)):     # processor_node.cc:66-74
        assert hip.dll.lio_pp_check_config(lo, up, rings, cfg) == 0
    cfg.max_corner_less_sharp, cfg.max_surf_flat = 60, 4
    assert hip.dll.lio_pp_check_config(-15.0, 15.0, 16, cfg) == 0
    cfg.max_surf_flat = 5
    assert hip.dll.lio_pp_check_config(-15.0, 15.0, 16, cfg) == -4
    assert oracle.dll.lio_pp_check_config(-15.0, 15.0, 16, cfg) == 0
    cfg.max_surf_flat, cfg.max_corner_sharp = 4, 61
    assert hip.dll.lio_pp_check_config(-15.0, 15.0, 16, cfg) == -1
    assert hip.dll.lio_pp_check_config(15.0, -15.0, 16, None) == -1


def test_process_imu_batch_equals_the_loop(oracle, hip):
    """lio_est_process_imu_batch = n calls of lio_est_process_imu (host-only in both libraries; the product's handle needs
    a device to exist, so without one only the oracle is exercised)."""
    import torch

    from lio_amd import pipeline

    libs = [oracle] + ([hip] if torch.cuda.is_available() else [])
    ds = synth.make_dataset("indoor", 7, 0.2, lidar=synth.Lidar(16, -15, 15, 450))
    for lib in libs:
        clouds = [pipeline.feature_clouds(lib, ds.lidar, f.scan)[0] for f in ds.frames]
        states = []
        for batch in (False, True):
            cfg = pipeline.config_indoor(lib, 4, 2)
            cfg.cutoff_deskew, cfg.keep_features, cfg.prior_factor = 1, 0, 1
            pipeline.set_extrinsic(cfg, ds)
            est = capi.Estimator(lib, cfg)
            pipeline.init_window(est, lib, ds, clouds, pos_sigma=0.005, rot_sigma=0.0005, vel_sigma=0.005)
            f = ds.frames[5]
            if batch:
                est.process_imu_batch(f.imu_dt, f.imu_acc, f.imu_gyr, f.imu_t)
            else:
                for j in range(f.imu_dt.shape[0]):
                    est.process_imu(float(f.imu_dt[j]), f.imu_acc[j], f.imu_gyr[j], float(f.imu_t[j]))
            states.append(est.get_window())
        for key in ("Ps", "Rs", "Vs"):
            np.testing.assert_array_equal(states[0][key], states[1][key])


def test_dropin_class_is_compiled_and_fails_loudly_without_a_gpu():
    """the drop-in for lio::Estimator (lio-mapping_amd/dropin/EstimatorHip.cc over the reference's MeasurementManager.cc,
    oracle/dropin_harness.cc) is a COMPILED binding: the library exists (rebuilt here when the reference tree is present), exports the
    harness entry points, and on a host without a GPU its constructor reports the failure instead of substituting anything."""
    import ctypes as C
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_ref", "libdropin_estimator.so")
    if os.path.isdir("/root/reference/src/imu_processor"):
        subprocess.run(["make", "-s", "-C", os.path.join(root, "lio-mapping_amd", "csrc")], check=True)
        subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "_ref/libdropin_estimator.so"], check=True)
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_estimator.so not built (needs the reference tree)")
    lib = C.CDLL(so)
    for sym in ("dropin_create", "dropin_destroy", "dropin_push_imu", "dropin_push_compact", "dropin_wait_processed", "dropin_get_stage",
                "dropin_get_window", "dropin_get_published"):
        assert hasattr(lib, sym), sym
    import torch

    if not torch.cuda.is_available():
        lib.dropin_create.restype = C.c_void_p
        lib.dropin_create.argtypes = [C.c_void_p] * 3 + [C.c_double] * 2
        ip = np.array([6, 3, 1, 1, 1, 1, 1, 0, 1, 1, 0, 1], np.int32)
        fp = np.array([0.2, 0.4, 1.0, 0.2, 0, 0, 0, 1, 0, 0, -0.08], np.float32)
        dp = np.array([0.1, 0.01, 0.0002, 2e-5, 9.805])
        assert lib.dropin_create(ip.ctypes.data, fp.ctypes.data, dp.ctypes.data, 0.0, 0.1) is None


def test_front_end_dropin_classes_are_compiled_and_fail_loudly_without_a_gpu():
    """SURVEY.md 8(b)'s other two classes: PointProcessorHip / PointOdometryHip (lio-mapping_amd/dropin, oracle/dropin_frontend.cc) compile against
    the reference's headers into oracle/_ref/libdropin_frontend.so, export the harness entry points, and without a GPU their constructors
    report the failure (no handle) instead of computing anything on the host."""
    import ctypes as C
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_ref", "libdropin_frontend.so")
    if os.path.isdir("/root/reference/src/point_processor"):
        subprocess.run(["make", "-s", "-C", os.path.join(root, "lio-mapping_amd", "csrc")], check=True)
        subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "_ref/libdropin_frontend.so"], check=True)
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdropin_frontend.so not built (needs the reference tree)")
    lib = C.CDLL(so)
    for pre, names in (("dropin_pp_", ("create", "destroy", "process", "count", "get", "ranges", "last_error")),
                       ("dropin_odom_", ("create", "destroy", "enable", "process", "get", "count", "get_cloud", "last_error"))):
        for n in names:
            assert hasattr(lib, pre + n), pre + n
    import torch

    if not torch.cuda.is_available():
        lib.dropin_pp_create.restype = C.c_void_p
        lib.dropin_pp_create.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        ic = np.array([8, 5, 2, 20, 4, 0], np.int32)
        dc = np.array([0.1, 0.2, 0.1, 0.2])
        assert lib.dropin_pp_create(-15.0, 15.0, 16, 0, ic.ctypes.data, dc.ctypes.data) is None
        lib.dropin_odom_create.restype = C.c_void_p
        lib.dropin_odom_create.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int]
        assert lib.dropin_odom_create(0.1, 2, 25, 0) is None
