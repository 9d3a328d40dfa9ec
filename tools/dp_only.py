#!/usr/bin/env python3
"""Profiling driver: a few device-resident DP plans, nothing else (no CPU
baseline, no iLQR).  usage: python3 tools/dp_only.py [grid] [steps] [blend order 0..3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

grid = sys.argv[1] if len(sys.argv) > 1 else "target"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dp, env, p, ep, path, occ = bench.dp_setup(grid, 0)
if len(sys.argv) > 3 and int(sys.argv[3]):
    dp.set_blend_order(int(sys.argv[3]))        # k_backward3o (rule R3 under another order / as the eight-corner sum)
for _ in range(steps):
    dp.plan_resident(env)
dp.sync()
print("done", grid, steps)
