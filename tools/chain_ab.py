#!/usr/bin/env python3
"""Chained sweep against stage-by-stage launches, same process, one plan at a time: p50 wall time of `update()` (host buffers in
and out) and microseconds per stage sweep (HIP events around a plan's sweeps / its stages) on grids that take k_backward3v.
usage: python3 tools/chain_ab.py [calls per leg]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpl_amd  # noqa: E402
from benchkit import scenarios as sc  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
print(f"{'grid':>18} {'form':>15} {'update() p50 ms':>16} {'us / stage sweep':>17}")
for S, V, A, T in [(256, 128, 64, 50), (256, 64, 64, 20), (256, 40, 52, 12), (128, 64, 66, 9), (512, 64, 32, 8), (512, 8, 6, 10), (512, 256, 128, 6)]:
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    env = tpl_amd.DynProgEnvironment(); env.reinit_buffers(ep); env.set_occ_map(sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC))
    dp = tpl_amd.DynProgLonPlanner(); dp.reinit_buffers(p)
    init = sc.initial_state(0.0, 8.0, 0.0)
    res = {}
    for rnd in range(2):
        for mode in (0, -1):
            dp.set_chained_sweep(mode)
            for _ in range(3):
                dp.update(init, path, env)
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter(); dp.update(init, path, env); ts.append(time.perf_counter() - t0)
            dp.timing_enable(True)
            for _ in range(4):
                dp.update(init, path, env)
            ms, n = dp.timing_get()
            dp.timing_enable(False)
            info = dp.info()
            res.setdefault(mode, []).append((float(np.median(ts)) * 1e3, ms / max(1, n) * 1e3, info["chained_sweep"], info["sweep"]))
    for mode in (0, -1):
        a = res[mode]
        name = ("chained" if a[0][2] else "not eligible") if mode else "stage by stage"
        print(f"{S}x{V}x{A}x{T:>3} {name:>15} {min(x[0] for x in a):>16.3f} {min(x[1] for x in a):>17.2f}   ({a[0][3]})", flush=True)
