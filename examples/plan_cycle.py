#!/usr/bin/env python3
"""One planning cycle through the five engines, chained the way tpl's PolyLatDpLonPlanner and
its trajectory-tracking controller chain them (poly_lat_dp_lon_planner.py:120-200,
model_predictive_controller.py): predictions -> occupancy -> lateral path -> DP (s,v,a) speed
plan -> MPC tracking, plus the (s, ds, l) planner on the same scene.  Needs an MI355X."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpl_amd                                      # noqa: E402
from benchkit import scenarios as sc                 # noqa: E402


def main():
    t0 = time.perf_counter()
    # --- environment: reference line + predicted objects -> Frenet occupancy and distance maps
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = sc.env_scene("arc", seed=1, n_objects=8)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(rl, step)
    for geom, stationary in objs:
        env.insert_geometry(geom, stationary)
    env.update()
    occ = env.get_occ_map()
    print(f"occupancy: {int((occ == 1).sum())} occupied, {int((occ == 3).sum())} stationary cells "
          f"of {occ.size}")

    # --- lateral plan
    lat = tpl_amd.PolyLatPlanner()
    lp = tpl_amd.PolyLatPlannerParams()
    lat.reinit_buffers(lp)
    start = tpl_amd.PolyLatTrajPoint()
    start.s, start.l, start.dl, start.v = 2.0, 0.3, 0.0, 8.0
    traj_lat = lat.update(start, env)
    print(f"lateral plan: quintic to l = {traj_lat.points[-1].l:+.2f} m, lattice cell {lat.best}")

    # --- longitudinal DP plan on the resampled path (PathState: x, y, s, l, k, v_max, distance)
    p = sc.lon_params(s_steps=201, v_steps=37, a_steps=20, t_steps=10)
    samples = traj_lat.lerp(np.arange(p.path_steps) * p.path_step_size)
    path = np.stack([samples[:, 7], samples[:, 8], samples[:, 5], samples[:, 1], samples[:, 11],
                     samples[:, 6], samples[:, 10]], axis=1).astype(np.float32)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    lon = dp.update(sc.initial_state(0.0, 8.0, 0.0), path, env)
    a = lon.as_numpy()
    print("speed plan  s:", np.round(a[:, 0], 1), " v:", np.round(a[:, 1], 1))

    # --- the (s, ds, l) planner on the same scene
    ll = tpl_amd.DynProgLatLonPlanner()
    s0 = tpl_amd.LatLonState()
    s0.s, s0.ds, s0.l = 2.0, 8.0, 0.3
    ll.traj_dp[0] = s0
    ll.traj_smooth[0] = s0
    ll.update_traj_dp(env)
    ll.update_traj_smooth(env)
    ll.update_traj_cart(env.ref_line)
    b = ll.traj_dp.as_numpy()
    print("(s,ds,l) plan s:", np.round(b[:, 1], 1), " l:", np.round(b[:, 5], 2))

    # --- MPC tracking of the speed plan start (batch of 1)
    from tpl_amd.optim import genopt, optimizers as o
    from benchkit import ilqr as bench_ilqr
    make = genopt.build(o.config_trajectory_tracking_mpc(), "trajectory_tracking_mpc")
    mpc = make(device=0)
    mpc.horizon = 20
    bench_ilqr._configure(mpc, np.array([0.0, 0.3, 0.0, 0.0, 8.0, 0.0, 0.0]))
    mpc.update()
    print(f"MPC: {int(np.ravel(mpc.iterations)[0])} iterations, first control {np.round(mpc.u[0], 3)}")
    print(f"cycle wall time incl. first-call set-up: {(time.perf_counter() - t0) * 1e3:.0f} ms")


if __name__ == "__main__":
    main()
