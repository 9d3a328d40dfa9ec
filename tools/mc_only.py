#!/usr/bin/env python3
"""Profiling driver: the Monte-Carlo scenario workload of bench.py (one pass over a reduced set) and nothing else.
usage: python3 tools/mc_only.py [scenarios]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import tpl_amd  # noqa: E402
from benchkit import scenarios as sc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S, V, A, T = sc.GRIDS["config2"]
p = sc.lon_params(S, V, A, T)
ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
occs, inits = bench.mc_scenarios(ep, 0, n)
dp = tpl_amd.DynProgLonPlanner(0)
dp.reinit_buffers(p)
dp.upload_scenarios(occs, ep, inits, sc.straight_path(p.path_steps, p.path_step_size), chunk=bench.MC_CHUNK)
for first in range(0, n, bench.MC_CHUNK):
    dp.plan_scenarios(first, min(bench.MC_CHUNK, n - first), id_base=0, reset_best=(first == 0))
print("done", dp.best_record()[:3])
