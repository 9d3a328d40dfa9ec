#!/usr/bin/env python3
"""Wall time of one plan (reinit + update, host in / host out) on a given grid: plan_time_grid.py S V A T [plans]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpl_amd                                       # noqa: E402
from benchkit import scenarios as sc                  # noqa: E402

S, V, A, T = (int(x) for x in sys.argv[1:5])
plans = int(sys.argv[5]) if len(sys.argv) > 5 else 40
p = sc.lon_params(S, V, A, T)
ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
path = sc.straight_path(p.path_steps, p.path_step_size)
env = tpl_amd.DynProgEnvironment(0)
env.reinit_buffers(ep)
env.set_occ_map(sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC))
dp = tpl_amd.DynProgLonPlanner(0)
init = sc.initial_state(0.0, 8.0, 0.0)
lat = []
for k in range(plans):
    p.dt_start = 1.0 - 0.01 * (k % 50)
    ep.dt_start = p.dt_start
    t0 = time.perf_counter()
    env.reinit_buffers(ep)
    dp.reinit_buffers(p)
    dp.update(init, path, env)
    lat.append((time.perf_counter() - t0) * 1e3)
info = dp.info()
print(f"{S}x{V}x{A} x {T}: p50 {np.percentile(lat[5:], 50):.3f} ms  = {np.percentile(lat[5:], 50) * 1e3 / (T - 1):.2f} us per stage incl. everything; kchunk {info['kchunk']} sweep {info['sweep']}")
