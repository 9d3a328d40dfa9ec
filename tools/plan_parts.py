#!/usr/bin/env python3
"""Host-side p50 of the three calls of a planning cycle on tpl's default grid: env.reinit_buffers, dp.reinit_buffers,
dp.update (1.7 / 1.7 / 124 us on the round-3 box; the kernels of update() span 112 us, tools/plan_timeline.py)."""
import time, numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpl_amd
from benchkit import scenarios as sc
S, V, A, T = sc.GRIDS["default"]
p = sc.lon_params(S, V, A, T)
ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
path = sc.straight_path(p.path_steps, p.path_step_size)
env = tpl_amd.DynProgEnvironment(0)
env.reinit_buffers(ep)
env.set_occ_map(sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC))
dp = tpl_amd.DynProgLonPlanner(0)
init = sc.initial_state(0.0, 8.0, 0.0)
a=[];b=[];c=[]
for k in range(300):
    p.dt_start = 1.0 - 0.01 * (k % 50); ep.dt_start = p.dt_start
    t0 = time.perf_counter(); env.reinit_buffers(ep)
    t1 = time.perf_counter(); dp.reinit_buffers(p)
    t2 = time.perf_counter(); dp.update(init, path, env)
    t3 = time.perf_counter()
    a.append(t1-t0); b.append(t2-t1); c.append(t3-t2)
f=lambda x: round(float(np.percentile(np.array(x[20:])*1e6,50)),1)
print("env.reinit", f(a), "dp.reinit", f(b), "update", f(c), "us p50")
