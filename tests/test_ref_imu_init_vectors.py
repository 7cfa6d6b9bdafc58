"""The oracle's AND the product's IMU initialisation against THE REFERENCE'S OWN ImuInitializer.cc.

tests/golden/ref_imu_init_vectors.npz holds what hyye/lio-mapping's src/imu_processor/ImuInitializer.cc — compiled where it lies
against the stand-ins of oracle/ref_shim (`make -C oracle ref`) — returns on the synthetic windows of tests/test_imu_init.py:
Initialization (gyro bias with the re-propagation of every interval, gravity approximation with its 1.0 m/s^2 acceptance band, five
rounds of the tangent-space refinement, R_WI) and EstimateExtrinsicRotation (Huber-weighted quaternion system, 0.25 acceptance).
Stood in: Eigen's dense API; A.ldlt().solve and JacobiSVD forwarded to the oracle's restatements (so those two are not
independently pinned); Sophus::SO3::exp.  Both libraries are loaded on the CPU: these entry points are host code in the product."""
import os
import sys

import numpy as np
import pytest

from lio_amd import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import test_imu_init as T  # noqa: E402

V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_imu_init_vectors.npz"))


def _cases():
    from lio_amd import synth
    return {"default": {}, "short": dict(n=5), "biased_long": dict(n=12, frame_dt=0.3, bg=(-0.01, 0.006, 0.002)),
            "extrinsic": dict(R_lb=synth.rot_zyx(0.4, -0.25, 0.3), bg=(0, 0, 0), traj=synth.Trajectory(ang_scale=3.0)),
            "extrinsic_weak": dict(R_lb=synth.rot_zyx(0.1, 0.05, -0.2), bg=(0, 0, 0))}


@pytest.fixture(scope="module", params=["oracle", "product"])
def lib(request, oracle):
    return oracle if request.param == "oracle" else capi.load_hip()


@pytest.mark.parametrize("name", ["default", "short", "biased_long"])
def test_initialization_matches_the_reference(lib, name):
    tr, pims, T_lb, _ = T._window(lib, **_cases()[name])
    r = lib.imu_initialization(tr, pims, T_lb)
    assert int(r["ok"]) == int(V[f"{name}_ok"])
    if not r["ok"]:
        return
    np.testing.assert_allclose(r["Bgs"], V[f"{name}_Bgs"], atol=1e-12)
    np.testing.assert_allclose(r["Vs"], V[f"{name}_Vs"], atol=1e-10)
    np.testing.assert_allclose(r["g"], V[f"{name}_g"], atol=1e-11)
    np.testing.assert_allclose(r["R_WI"], V[f"{name}_R"], atol=1e-12)


@pytest.mark.parametrize("name", ["extrinsic", "extrinsic_weak"])
def test_extrinsic_rotation_matches_the_reference(lib, name):
    tr, pims, T_lb, _ = T._window(lib, **_cases()[name])
    ok, q = lib.imu_estimate_extrinsic_rotation(tr, pims, ([0, 0, 0, 1], T_lb[1]))
    assert int(ok) == int(V[f"{name}_ok"])
    qr = V[f"{name}_q"]
    assert min(np.abs(q - qr).max(), np.abs(q + qr).max()) < 1e-6          # float32 quaternion in the transform


def test_committed_vectors_are_what_the_reference_produces(tmp_path):
    """Build container only."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/src/imu_processor"):
        pytest.skip("the reference tree is not on this machine")
    subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"], check=True)
    gen = os.path.join(root, "tests", "golden", "make_ref_imu_init_vectors.py")
    out = str(tmp_path / "v.npz")
    code = open(gen).read().replace('path = os.path.join(HERE, "ref_imu_init_vectors.npz")', f"path = {out!r}").replace("__file__", repr(gen))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True)
    fresh = np.load(out)
    for k in V.files:
        np.testing.assert_array_equal(fresh[k], V[k], err_msg=k)
