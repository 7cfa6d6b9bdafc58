"""Conditioning of a ProcessLaserOdom chain, measured on the CPU oracle against ITSELF (no GPU involved).

Two oracle estimators run the same HDL-64E window-15 / opt-window-5 chain.  Before every step estimator A receives
estimator B's window states plus an N(0, eps) perturbation of the positions (eps = 1e-8 m by default), and
  mode "states"  : nothing else (A keeps its own marginalization prior and extrinsic),
  mode "all"     : also B's prior and extrinsic (tests/window_util.force_all).
The printed gap after each step is what a 1e-8 m difference of the INPUTS turns into — the floor below which a
step-by-step comparison of two implementations cannot be read as an implementation difference.

  python tests/golden/make_chain_sensitivity.py outdoor 12 states > tests/golden/chain_sensitivity_states.txt
  python tests/golden/make_chain_sensitivity.py outdoor 12 all    > tests/golden/chain_sensitivity_all.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from lio_amd import capi, pipeline  # noqa: E402
from window_util import force_all, force_window, make_pair, window_gap  # noqa: E402

kind, n_chain, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
eps = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-8
W, Wo, dt = 15, 5, 0.3 if kind == "outdoor" else 0.2
orc = capi.LioLib(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
ds, clouds, (ea, eb) = make_pair((orc, orc), kind, W, Wo, W + 1 + n_chain, dt)
for e in (ea, eb):
    e.solve()
    e.slide()
print(f"# {kind} window {W}/{Wo}, eps = {eps:g} m on the positions handed to A before every step, mode = {mode}")
print("# step  max|dP| m   max rot gap rad   |dJtJ|/max of the priors   lidar factors A/B")
for k in range(W + 1, W + 1 + n_chain):
    if mode == "all":
        force_all(ea, eb, ds)
    w = dict(eb.get_window())
    w["Ps"] = w["Ps"] + eps * np.random.default_rng(k).normal(size=w["Ps"].shape)
    force_window(ea, w, ds)
    ra = pipeline.feed_frame(ea, ds, k, clouds[k][0], clouds[k][1])
    rb = pipeline.feed_frame(eb, ds, k, clouds[k][0], clouds[k][1])
    g = window_gap(ea.get_window(), eb.get_window())
    pa, pb = ea.prior(), eb.prior()
    rel = np.max(np.abs(pa["JtJ"] - pb["JtJ"])) / np.abs(pb["JtJ"]).max()
    print(f"{k:4d}   {g[0]:.2e}    {g[1]:.2e}          {rel:.1e}                  {ra.n_lidar_residuals}/{rb.n_lidar_residuals}")
