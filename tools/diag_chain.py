"""Diagnostic (not a test): per-step gaps between the product and the oracle on teacher-forced chains."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
from lio_amd import capi, pipeline
from window_util import make_pair, window_gap, force_window

hip = capi.load_hip(); orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))

def chain(kind, W, Wo, n_chain, dt, keep, deskew, pf, force=True):
    ds, clouds, (ea, eb) = make_pair((hip, orc), kind, W, Wo, W + 1 + n_chain, dt, keep=keep, deskew=deskew, prior_factor=pf, pp_lib=hip)
    ra, rb = ea.solve(), eb.solve()
    print(f"== {kind} W{W}/Wo{Wo} keep={keep} deskew={deskew} pf={pf} force={force}")
    print("first", window_gap(ea.get_window(), eb.get_window())[:2], ra.iterations, rb.iterations, ra.n_lidar_residuals - rb.n_lidar_residuals)
    for e in (ea, eb): e.slide()
    if force: force_window(ea, eb.get_window(), ds)
    for k in range(W + 1, W + 1 + n_chain):
        ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1]); rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
        wa, wb = ea.get_window(), eb.get_window()
        g = window_gap(wa, wb)
        n = rb.iterations + 1
        ta, tb = np.array(ra.cost_trace[:n]), np.array(rb.cost_trace[:n])
        pg = ra.cost_marg_before - rb.cost_marg_before
        print(f"k={k} dP {g[0]:.2e} dR {g[1]:.2e} dV {g[2]:.2e} it {ra.iterations}/{rb.iterations} term {ra.termination}/{rb.termination} succ {ra.successful_steps}/{rb.successful_steps} "
              f"lo {ra.laser_odom_iterations}/{rb.laser_odom_iterations} nres {ra.n_lidar_residuals}/{rb.n_lidar_residuals} conv {ra.convergence_flag}/{rb.convergence_flag} "
              f"dext {np.max(np.abs(wa['t_lb']-wb['t_lb'])):.1e} {np.max(np.abs(wa['q_lb']-wb['q_lb'])):.1e} priorgap {pg:.2e} trace rel {np.max(np.abs(ta-pg-tb)/tb):.2e} cost0 {tb[0]:.4g} costN {tb[-1]:.6g} margcost {rb.cost_marg_before:.3g}")
        if force: force_window(ea, wb, ds)

chain("indoor", 15, 5, 6, 0.2, 0, False, 1)
chain("indoor", 12, 7, 6, 0.2, 1, True, 1)
chain("indoor", 12, 7, 6, 0.2, 1, True, 0)
chain("outdoor", 15, 5, 20, 0.3, 0, False, 1)
