#!/usr/bin/env python3
"""k_ll_backward's rate against the grid size: cell-actions/s of DynProgLatLonPlanner.update_traj_dp's backward stages
(49 actions per cell) on tpl's default 201x37x21 grid -- 2 442 wavefronts, 2.4 per SIMD: the chip is not filled -- and on
grids that fill it, by HIP events around the stage launches (DynProgLatLonPlanner.timing)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpl_amd                                       # noqa: E402
from benchkit import latency                         # noqa: E402

ep, rl, step, objs, env, _ = latency.poly_lat_scene(0)
rows = []
for S, DS, L in ((201, 37, 21), (201, 74, 41), (401, 74, 41), (401, 148, 81)):
    p = tpl_amd.DynProgLatLonPlannerParams()
    p.length_veh, p.width_veh = 4.9, 1.9
    p.s_steps, p.ds_steps, p.l_steps = S, DS, L
    pl = tpl_amd.DynProgLatLonPlanner(0)
    start = tpl_amd.LatLonState()
    start.s, start.ds, start.l = 2.0, 8.0, 0.3
    lat = []
    for k in range(12):
        pl.reinit_buffers(p)
        pl.traj_dp[0] = start
        t0 = time.perf_counter()
        pl.update_traj_dp(env)
        lat.append(time.perf_counter() - t0)
    t = float(np.median(lat[3:]))
    cells = S * DS * L
    rows.append({"grid": [S, DS, L], "wavefronts": (S * DS + 255) // 256 * 4 * L, "update_traj_dp_ms": t * 1e3,
                 "cell_actions_per_s_whole_call": cells * 49 * (p.t_steps - 1) / t})
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps(rows, indent=1))
