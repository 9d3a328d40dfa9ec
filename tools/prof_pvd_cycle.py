#!/usr/bin/env python3
"""cProfile over 60 cycles of PathVelDecompPlanner.update on the HIP solvers (where the host side of tplsim's default
planner spends its time)."""
import cProfile
import os
import pstats
import sys
from contextlib import contextmanager

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pvd_cases as cases                                                   # noqa: E402
from tpl_amd.planning.path_vel_decomp import PathVelDecompPlanner           # noqa: E402
from tpl_amd.store import Bundle                                            # noqa: E402


@contextmanager
def lock():
    yield


sh = Bundle()
pl = PathVelDecompPlanner(sh, lock)
for kind in ("path_optim", "velocity_optim", "path_smoothing"):
    for k, v in vars(cases.default_params(kind)).items():
        setattr(getattr(sh.params, kind), k, v)
envs = [cases.shared_environment(c % 12) for c in range(70)]
for e in envs[:10]:
    pl.update(e)
pr = cProfile.Profile()
pr.enable()
for e in envs[10:]:
    pl.update(e)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
