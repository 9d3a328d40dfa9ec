#!/usr/bin/env python3
"""Kernel time of one single-problem update() against the iteration cap: fixed cost per call vs cost per iteration."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilqr_cases as cases                                  # noqa: E402
from tpl_amd.optim import genopt, optimizers as o           # noqa: E402

for model, setup in (("velocity_profile_space", lambda opt: cases.velocity_space(opt, seed=0, horizon=250)),
                     ("lateral_profile", lambda opt: cases.lateral(opt, seed=0, horizon=250)),
                     ("trajectory_tracking_mpc", lambda opt: cases.mpc(opt, seed=0, horizon=20, min_rel=0.0))):
    make = genopt.build(getattr(o, "config_" + model)(), model)
    opt = make()
    rows = []
    for cap in (1, 2, 4, 8, 16):
        ks = []
        for rep in range(12):
            setup(opt)
            opt.max_iterations = cap
            opt.min_rel_cost_change = 0.0
            opt.timing(reset=True)
            opt.update()
            k_ms, n = opt.timing(reset=True)
            ks.append(k_ms / max(1, n))
        rows.append((cap, int(np.ravel(opt.iterations)[0]), float(np.median(ks[3:]))))
    its = np.array([r[1] for r in rows], float)
    ms = np.array([r[2] for r in rows])
    slope, icpt = np.polyfit(its, ms, 1)
    print(f"{model:26s}", [(r[1], round(r[2], 3)) for r in rows], f" -> {icpt:.3f} ms fixed + {slope:.3f} ms per iteration")
