"""CPU checks of the oracle's scan-to-scan odometry (no GPU): it recovers the ground-truth sweep motion of the
synthetic motion-distorted sweeps, and the stored clouds are the TransformToEnd images of the inputs."""
import numpy as np

from lio_amd import capi, synth


def test_oracle_odometry_recovers_motion(oracle):
    sweeps, pose_fn, lid = synth.make_sweeps("indoor", 3)
    od = capi.PointOdometry(oracle, 0.1, 2, 25, False)
    for k, sw in enumerate(sweeps):
        pp = capi.PointProcessor(oracle, lid.lower_deg, lid.upper_deg, lid.rings)
        pp.process(sw)
        r = od.process(pp.cloud(1), pp.cloud(2), pp.cloud(3), pp.cloud(4))
        if k == 0:
            assert r["iterations"] == 0  # first sweep only initialises (PointOdometry.cc:302-310)
            continue
        R0, p0 = pose_fn(1.0 + 0.1 * k)
        R1, p1 = pose_fn(1.0 + 0.1 * (k + 1))
        t_gt = R1.T @ (p0 - p1)   # T_{end<-start}
        assert r["num_selected"] > 300
        assert np.linalg.norm(r["T_es"][1] - t_gt) < 0.2   # 0.6 m of motion per sweep recovered to < 0.2 m (damped GN, 25 its)
        assert abs(np.linalg.norm(r["T_es"][0]) - 1.0) < 1e-5  # normalised after the sweep (:663)
        lc = od.last_cloud(1)
        assert np.all(lc[:, 3] == np.floor(lc[:, 3]))  # TransformToEnd strips the relative time (:276)
