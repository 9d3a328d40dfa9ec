#!/usr/bin/env python3
"""cProfile of the controller plugin's tick on the HIP solver (where the host time of an MPC tick goes)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mpc_cases as cases                                   # noqa: E402
from tpl_amd.control import ModelPredictiveController       # noqa: E402

cases.CASES["prof"] = (dict(x=0.0, y=0.5, yaw=0.08, v=6.0), dict(horizon=20), 400)
pr = cProfile.Profile()
cases.run(lambda sh, lock: ModelPredictiveController(sh, lock), "default")        # warm-up
pr.enable()
cases.run(lambda sh, lock: ModelPredictiveController(sh, lock), "prof")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
