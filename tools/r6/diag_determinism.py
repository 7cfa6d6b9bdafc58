"""B copies of the headline window (HDL-64E, window 15 / opt 5) through lio_est_batch: per stage, which copies differ from copy 0
(lio_est_batch_stage_digest).  usage: diag_determinism.py B trials [steps_per_trial]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
import torch  # noqa: F401,E402  (torch first: see __graft_entry__.py)
from lio_amd import capi, pipeline, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hip = capi.load_hip()
W, Wo = 15, 5
ds = synth.make_dataset("outdoor", W + 2, 0.3)
clouds = [pipeline.feature_clouds(hip, ds.lidar, f.scan) for f in ds.frames]
cfg = pipeline.config_outdoor64(hip, W, Wo)
cfg.keep_features, cfg.prior_factor, cfg.cutoff_deskew, cfg.opt_extrinsic = 0, 1, 1, 0
pipeline.set_extrinsic(cfg, ds)
est0 = capi.Estimator(hip, cfg)
pipeline.init_window(est0, hip, ds, [c[0] for c in clouds], pos_sigma=0.01, rot_sigma=0.001, vel_sigma=0.01, seed=3)
est0.snapshot()
clones = []
for _ in range(B):
    e = capi.Estimator(hip, cfg)
    e.copy_snapshot_of(est0)
    e.restore()
    clones.append(e)
batch = capi.EstimatorBatch(hip, clones)
env = {k: v for k, v in os.environ.items() if k.startswith("LIO_")}
print(f"B {B} trials {trials} steps {steps} env {env}", flush=True)
bad_total = 0
for t in range(trials):
    reps = batch.solve_restored(steps)
    keys = [(r.iterations, r.successful_steps, r.termination, r.n_lidar_residuals, r.n_local_map, r.laser_odom_iterations, r.final_cost) for r in reps]
    line = [f"trial {t}: rep0 {keys[0]}"]
    for s, name in enumerate(capi.EstimatorBatch.STAGES):
        d = batch.stage_digest(s)
        bad = np.nonzero(d != d[0])[0]
        if bad.size:
            line.append(f"{name}: {bad.size} differ {bad[:12].tolist()}")
            bad_total += 1
    badk = [w for w in range(B) if keys[w] != keys[0]]
    if badk:
        line.append(f"reports differ at {badk[:12]}: {[keys[w] for w in badk[:3]]}")
    print(" | ".join(line), flush=True)
if os.environ.get("LIO_DEBUG_DIGEST"):
    batch.stage_digest(8)
    batch.stage_digest(9)
print("RESULT", "deterministic" if bad_total == 0 else f"{bad_total} stage mismatches")
