"""The other two COMPILED drop-ins of SURVEY.md 8(b): lio-mapping_amd/dropin/PointProcessorHip.{h,cc} and PointOdometryHip.{h,cc} — classes with
lio::PointProcessor's / lio::PointOdometry's public surface over include/lio_c.h — built against the reference's headers
(oracle/dropin_frontend.cc -> oracle/_ref/libdropin_frontend.so, `make -C oracle ref`) and driven with the SAME calls as the reference's
own classes compiled from PointProcessor.cc / PointOdometry.cc (oracle/_ref/libref_pointproc.so, libref_odometry.so):

* PointProcessorHip through the ROS-free sequence of test_point_processor.cc:103-106 (SetInputCloud -> PointToRing ->
  ExtractFeaturePoints): the public members laser_scans / scan_ranges and the four feature clouds against the reference's, points and
  order bit-exact (the relative time inside the intensity within atan2f's last ulp, as tests/test_gpu_parity.py bounds it);
  intensity_scans against the full-resolution cloud the reference publishes (cloud_in_rings_, same values in ring order);
* PointOdometryHip through its five message handlers + Process(), handed the reference's own feature clouds and /full_cloud:
  transform_es_ (1e-5), transform_sum_ (1e-4), the clouds kept for the next sweep and the published
  /compact_data message (header, sizes, every point) against the reference's."""
import ctypes as C
import os

import numpy as np
import pytest

import ref_odom_cases as oc
from lio_amd import synth
from ref_pp_cases import cases as pp_cases

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(HERE, "..", "oracle", "_ref")
fp = C.POINTER(C.c_float)


def _so(name):
    path = os.path.join(REFDIR, name)
    assert os.path.exists(path), f"oracle/_ref/{name} is missing: run `make -C oracle ref` where /root/reference exists (build())"
    return C.CDLL(path)


def _declare_pp(lib, pre):
    f = lambda n: getattr(lib, pre + n)   # noqa: E731
    f("create").restype = C.c_void_p
    f("create").argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f("destroy").argtypes = [C.c_void_p]
    f("process").argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    f("count").restype = C.c_size_t
    f("count").argtypes = [C.c_void_p, C.c_int]
    f("get").argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    f("ranges").argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return f


def _declare_odom(lib, pre):
    f = lambda n: getattr(lib, pre + n)   # noqa: E731
    f("create").restype = C.c_void_p
    f("create").argtypes = [C.c_float, C.c_int, C.c_int, C.c_int]
    f("destroy").argtypes = [C.c_void_p]
    f("enable").argtypes = [C.c_void_p, C.c_int]
    f("process").argtypes = [C.c_void_p] + [fp, C.c_size_t] * 5 + [C.c_double]
    f("get").argtypes = [C.c_void_p, fp, fp, C.POINTER(C.c_long)]
    f("count").restype = C.c_size_t
    f("count").argtypes = [C.c_void_p, C.c_int]
    f("get_cloud").argtypes = [C.c_void_p, C.c_int, fp]
    return f


def _cloud(f, h, which, getter="get"):
    n = f("count")(h, which)
    a = np.zeros((n, 4), np.float32)
    if n:
        f(getter)(h, which, a.ctypes.data_as(fp))
    return a


def _pp_pair(lid, over):
    ref, drop = _declare_pp(_so("libref_pointproc.so"), "ref_pp_"), _declare_pp(_so("libdropin_frontend.so"), "dropin_pp_")
    cfg = {"num_scan_subregions": 8, "num_curvature_regions": 5, "max_corner_sharp": 2, "max_corner_less_sharp": 20, "max_surf_flat": 4, "infer_start_ori": 0,
           "surf_curv_th": 0.1, "less_flat_filter_size": 0.2, "scan_period": 0.1, "rad_diff": 0.2}
    cfg.update({k: v for k, v in over.items() if k != "uneven"})
    ic = np.array([cfg[k] for k in ("num_scan_subregions", "num_curvature_regions", "max_corner_sharp", "max_corner_less_sharp", "max_surf_flat", "infer_start_ori")], np.int32)
    dc = np.array([cfg[k] for k in ("surf_curv_th", "less_flat_filter_size", "scan_period", "rad_diff")], np.float64)
    hs = []
    for f in (ref, drop):
        h = f("create")(lid.lower_deg, lid.upper_deg, lid.rings, int(over.get("uneven", 0)), ic.ctypes.data, dc.ctypes.data)
        assert h, "PointProcessorHip could not create its library handle (no GPU?)"
        hs.append(h)
    return ref, drop, hs


PP_CASES = [c for c in pp_cases() if c[0] != "indoor_infer_start_ori"]   # (points on the inferred seam wrap by one period: tests/test_gpu_parity.py has that case)


def test_config_defaults_are_the_reference_s():
    """the dict above restates PointProcessorConfig's defaults (PointProcessor.h:107-124), which lio_pp_default_config returns too"""
    from lio_amd import capi
    lib = capi.LioLib(os.path.join(HERE, "..", "lio-mapping_amd", "csrc", "liblio_hip.so"))
    c = capi.PPConfig()
    lib.dll.lio_pp_default_config(c)
    assert (c.num_scan_subregions, c.num_curvature_regions, c.max_corner_sharp, c.max_corner_less_sharp, c.max_surf_flat) == (8, 5, 2, 20, 4)
    assert abs(c.surf_curv_th - 0.1) < 1e-7 and abs(c.less_flat_filter_size - 0.2) < 1e-7 and abs(c.rad_diff - 0.2) < 1e-12 and abs(c.scan_period - 0.1) < 1e-12


@pytest.mark.parametrize("case", PP_CASES, ids=[c[0] for c in PP_CASES])
def test_point_processor_class_against_the_reference_s(case):
    name, lid, over, sweeps = case
    ref, drop, (hr, hd) = _pp_pair(lid, over)
    worst = 0.0
    for k, (scan, ring) in enumerate(sweeps):
        scan = np.ascontiguousarray(scan, np.float32)
        rp = None if ring is None else np.ascontiguousarray(ring, np.uint16)
        for f, h in ((ref, hr), (drop, hd)):
            f("process")(h, scan.ctypes.data, len(scan), None if rp is None else rp.ctypes.data)
        ra, rb = np.zeros((lid.rings, 2), np.int64), np.zeros((lid.rings, 2), np.int64)
        ref("ranges")(hr, lid.rings, ra.ctypes.data)
        drop("ranges")(hd, lid.rings, rb.ctypes.data)
        np.testing.assert_array_equal(ra, rb, err_msg=f"{name} sweep {k}: scan_ranges")
        for w in (5, 1, 2, 3, 4, 0):          # laser_scans, the four feature clouds, the full-resolution cloud
            a, b = _cloud(ref, hr, w), _cloud(drop, hd, w)
            assert a.shape == b.shape and len(a) > 0, (name, k, w, a.shape, b.shape)
            np.testing.assert_array_equal(a[:, :3], b[:, :3], err_msg=f"{name} sweep {k} cloud {w}")
            np.testing.assert_allclose(a[:, 3], b[:, 3], atol=8e-6)
            worst = max(worst, float(np.abs(a[:, 3] - b[:, 3]).max()))
        full, inten = _cloud(ref, hr, 0), _cloud(drop, hd, 6)     # intensity_scans (public) hold what cloud_in_rings_ is built from (:193-201)
        assert full.shape == inten.shape
        np.testing.assert_array_equal(full[:, :3], inten[:, :3])
        np.testing.assert_allclose(full[:, 3], inten[:, 3], atol=8e-6)
    print(f"{name}: PointProcessorHip == PointProcessor.cc over {len(sweeps)} sweeps (points, order, ranges bit-exact; worst intensity gap {worst:.1e})")
    ref("destroy")(hr)
    drop("destroy")(hd)


@pytest.mark.parametrize("name", ["indoor_io2", "outdoor_io3", "indoor_no_deskew", "indoor_packer_after_1"])
def test_point_odometry_class_against_the_reference_s(name):
    case = [c for c in oc.cases() if c[0] == name][0]
    _, kind, n, io, no_deskew, disable_after = case
    sweeps, _, lid = synth.make_sweeps(kind, n)
    ref, drop = _declare_odom(_so("libref_odometry.so"), "ref_odom_"), _declare_odom(_so("libdropin_frontend.so"), "dropin_odom_")
    hr, hd = ref("create")(0.1, io, 25, no_deskew), drop("create")(0.1, io, 25, no_deskew)
    assert hd, "PointOdometryHip could not create its library handle (no GPU?)"
    rpp, _, (hp, hp_drop) = _pp_pair(lid, {})
    worst = {"T_es": 0.0, "T_sum": 0.0, "points": 0.0}
    published = 0
    for k, sw in enumerate(sweeps):
        sw = np.ascontiguousarray(sw, np.float32)
        rpp("process")(hp, sw.ctypes.data, len(sw), None)                      # the reference's own processor node in front of both
        clouds = [_cloud(rpp, hp, w) for w in (1, 2, 3, 4, 0)]                 # sharp, less sharp, flat, less flat, /full_cloud
        if disable_after is not None and k == disable_after:
            ref("enable")(hr, 0)
            drop("enable")(hd, 0)
        args = []
        for c in clouds:
            args += [c.ctypes.data_as(fp), len(c)]
        got = []
        for f, h in ((ref, hr), (drop, hd)):
            f("process")(h, *args, 1.0 + 0.1 * k)
            Te, Ts, fc = np.zeros(7, np.float32), np.zeros(7, np.float32), C.c_long(0)
            f("get")(h, Te.ctypes.data_as(fp), Ts.ctypes.data_as(fp), C.byref(fc))
            got.append((Te.astype(float), Ts.astype(float), fc.value, _cloud(f, h, 0, "get_cloud"), _cloud(f, h, 1, "get_cloud"), _cloud(f, h, 2, "get_cloud")))
        (Te_r, Ts_r, fc_r, corner_r, surf_r, comp_r), (Te_d, Ts_d, fc_d, corner_d, surf_d, comp_d) = got
        assert fc_r == fc_d
        np.testing.assert_allclose(Te_d, Te_r, atol=1e-5)
        np.testing.assert_allclose(Ts_d[4:], Ts_r[4:], atol=1e-4)
        assert min(np.abs(Ts_d[:4] - Ts_r[:4]).max(), np.abs(Ts_d[:4] + Ts_r[:4]).max()) < 1e-4
        worst["T_es"] = max(worst["T_es"], float(np.abs(Te_d - Te_r).max()))
        worst["T_sum"] = max(worst["T_sum"], float(np.abs(Ts_d[4:] - Ts_r[4:]).max()))

        def close(a, b, what):
            assert a.shape == b.shape, (name, k, what, a.shape, b.shape)
            if not len(a):
                return
            reach = np.linalg.norm(a[:, :3], axis=1)
            gap = np.abs(a[:, :3] - b[:, :3]).max(axis=1)
            assert (gap <= 1e-4 + 4e-5 * reach).all(), (name, k, what, float(gap.max()))   # T_es agrees to 1e-5: a lever arm of `reach`
            np.testing.assert_array_equal(a[:, 3], b[:, 3])                                   # int(intensity): the ring
            worst["points"] = max(worst["points"], float(gap.max()))

        close(corner_d, corner_r, "last_corner_cloud_")
        close(surf_d, surf_r, "last_surf_cloud_")
        assert len(comp_r) == len(comp_d), (name, k, len(comp_r), len(comp_d))     # published on the same sweeps (io_ratio), same size
        if len(comp_r):
            published += 1
            np.testing.assert_allclose(comp_d[0, :3], comp_r[0, :3], atol=1e-4)   # transform_sum_.pos
            assert min(np.abs(comp_d[1] - comp_r[1]).max(), np.abs(comp_d[1] + comp_r[1]).max()) < 1e-4
            np.testing.assert_array_equal(comp_d[2, :3], comp_r[2, :3])           # the three sizes
            sizes = comp_r[2, :3].astype(int)
            assert sizes.sum() + 3 == len(comp_r) and sizes[2] == len(clouds[4])
            body_r, body_d = comp_r[3:], comp_d[3:]
            if disable_after is not None and k >= disable_after:
                np.testing.assert_array_equal(body_d, body_r)                     # packer mode: everything passes through untouched
            else:
                close(body_d[:sizes[0] + sizes[1]], body_r[:sizes[0] + sizes[1]], "/compact_data features")
                close(body_d[sizes[0] + sizes[1]:], body_r[sizes[0] + sizes[1]:], "/compact_data full cloud (host TransformToEnd)")
    assert published >= 1
    print(f"{name}: PointOdometryHip vs PointOdometry.cc over {n} sweeps: worst |dT_es| {worst['T_es']:.1e}, |dT_sum.pos| {worst['T_sum']:.1e}, "
          f"point gap {worst['points']:.1e} m; {published} /compact_data messages compared")
    for f, h in ((ref, hr), (drop, hd)):
        f("destroy")(h)
    rpp("destroy")(hp)
