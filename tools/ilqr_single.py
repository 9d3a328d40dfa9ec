#!/usr/bin/env python3
"""Profiling driver: ONE small problem per call (lateral_profile, T = 250, 16 iterations), three updates."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilqr_cases as cases                                  # noqa: E402
from tpl_amd.optim import genopt, optimizers as o           # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "lateral_profile"
make = genopt.build(getattr(o, "config_" + model)(), model)
opt = make()
for _ in range(3):
    (cases.lateral if model == "lateral_profile" else cases.velocity_space)(opt, seed=0, horizon=250)
    opt.max_iterations = 16
    opt.min_rel_cost_change = 0.0
    opt.update()
print("done", int(opt.iterations))
