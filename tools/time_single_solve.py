#!/usr/bin/env python3
"""One problem per call: wall time of Optim.update() vs the kernel time inside it, per tpl problem as its caller runs it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilqr_cases as cases                                  # noqa: E402
from tpl_amd.optim import genopt, optimizers as o           # noqa: E402

PLANS = [("lateral_profile", lambda opt: cases.lateral(opt, seed=0, horizon=250)),
         ("velocity_profile_space", lambda opt: cases.velocity_space(opt, seed=0, horizon=250)),
         ("ref_line_smoother_k", lambda opt: cases.smoother(opt, seed=0, horizon=250)),
         ("trajectory_tracking_mpc", lambda opt: cases.mpc(opt, seed=0, horizon=20))]
for model, setup in PLANS:
    make = genopt.build(getattr(o, "config_" + model)(), model)
    opt = make()
    wall, its = [], []
    for rep in range(30):
        setup(opt)                                          # cold start every time: the same work per call
        opt.timing(reset=True)
        t0 = time.perf_counter()
        opt.update()
        wall.append((time.perf_counter() - t0) * 1e3)
        its.append(int(np.ravel(opt.iterations)[0]))
    k_ms, launches = opt.timing(reset=True)
    T = int(opt.horizon)
    print(f"{model:28s} T={T:3d}  update() p50 {np.percentile(wall[5:], 50):.3f} ms   kernel {k_ms / max(1, launches):.3f} ms   "
          f"iterations {its[-1]}  -> {1e3 * k_ms / max(1, launches) / max(1, its[-1]) / T:.2f} us per stage-iteration")
