"""Diagnostic (not a test): per-step component costs of the product and the oracle on a fully teacher-forced chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
from lio_amd import capi, pipeline
from window_util import make_pair, window_gap, force_all

hip = capi.load_hip(); orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))

def chain(kind, W, Wo, n_chain, dt, keep, deskew, pf):
    ds, clouds, (ea, eb) = make_pair((hip, orc), kind, W, Wo, W + 1 + n_chain, dt, keep=keep, deskew=deskew, prior_factor=pf)
    ra, rb = ea.solve(), eb.solve()
    print(f"== {kind} W{W}/Wo{Wo} keep={keep} deskew={deskew} pf={pf}")
    for e in (ea, eb): e.slide()
    for k in range(W + 1, W + 1 + n_chain):
        force_all(ea, eb, ds)
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1]); rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        wa, wb = ea.get_window(), eb.get_window()
        g = window_gap(wa, wb)
        sa, sb = ea.get_surf_stack(W - 1), eb.get_surf_stack(W - 1)   # the frame pushed in this step (after the slide it sits at W-1)
        dstack = np.max(np.abs(sa[:, :3] - sb[:, :3])) if sa.shape == sb.shape else -1
        print(f"k={k} dP {g[0]:.2e} it {ra.iterations}/{rb.iterations} lo {ra.laser_odom_iterations}/{rb.laser_odom_iterations} nres {ra.n_lidar_residuals}/{rb.n_lidar_residuals} "
              f"pim {ra.cost_pim_before:.9g}/{rb.cost_pim_before:.9g} ppp {ra.cost_ppp_before:.9g}/{rb.cost_ppp_before:.9g} marg {ra.cost_marg_before:.9g}/{rb.cost_marg_before:.9g} "
              f"t0 {ra.cost_trace[0]:.9g}/{rb.cost_trace[0]:.9g} newest-stack gap {dstack:.2e} n {sa.shape[0]}/{sb.shape[0]}")

chain("indoor", 12, 7, 6, 0.2, 1, True, 0)
chain("indoor", 12, 7, 3, 0.2, 1, True, 1)
