"""Plan-latency measurement through the drop-in surface (BASELINE configs[0],
"p50 plan latency on default scenario"): host buffers in, host buffers out, one
vehicle, exactly how tpl's planner / controller plugins would call the engines.

* MPC: the closed loop of benchkit/minisim.py -- the simulator's ego model (core.py:91-135, with
  actuator dead time) under the controller's per-tick use of ``trajectory_tracking_mpc`` with N = 20
  (the template's default horizon, optim.c:1900), tracking a straight 8 m/s plan, one ``Optim.update()``
  per 10 ms tick like tplsim's fixed_time_step, 1 000 ticks.
* DP: ``DynProgLonPlanner.update`` on the reference's default grid
  (201 x 37 x 20, 10 stages; data/params/planning/default/state.json).
"""
import time

import numpy as np


def mpc_closed_loop(device=0, ticks=1000, horizon=20):
    """BASELINE configs[0] reduced to the hot path (benchkit/minisim.py): the simulator's ego model with 0.1 s actuator
    dead time under the model-predictive controller's per-tick solver use, N = 20, 1 000 ticks of 10 ms."""
    from . import minisim
    from tpl_amd.optim import genopt, optimizers as o
    make = genopt.build(o.config_trajectory_tracking_mpc(), "trajectory_tracking_mpc")
    opt = make(device=device)
    opt.horizon = horizon
    controls, trace, con = minisim.closed_loop(opt, ticks=ticks)
    lat = np.array(con.solve_ms[10:])
    tick = np.array(con.tick_ms[10:])
    return {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
            "controller_tick_p50_ms": float(np.percentile(tick, 50)), "controller_tick_p99_ms": float(np.percentile(tick, 99)),
            "mean_iterations": float(np.mean(opt.iterations)), "ticks": int(len(lat)),
            "final_lateral_error_m": float(abs(trace[-1, 1])), "final_speed_mps": float(trace[-1, 3]),
            "what": f"Optim.update() of trajectory_tracking_mpc, batch 1, N={horizon}, HEUN, <=20 iterations, inside the "
                    "closed loop of benchkit/minisim.py (ego model with 0.1 s dead time under the ModelPredictiveController plugin, straight 8 m/s plan); "
                    "host mirrors in and out (H2D + kernel + D2H)"}


def dp_default_plan(device=0, plans=60):
    import tpl_amd
    from . import scenarios as sc
    S, V, A, T = sc.GRIDS["default"]
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    env = tpl_amd.DynProgEnvironment(device)
    env.reinit_buffers(ep)
    env.set_occ_map(sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC))
    dp = tpl_amd.DynProgLonPlanner(device)
    lat = []
    init = sc.initial_state(0.0, 8.0, 0.0)
    for k in range(plans):
        p.dt_start = 1.0 - 0.01 * (k % 50)          # the caller changes dt_start every cycle
        ep.dt_start = p.dt_start
        t0 = time.perf_counter()
        env.reinit_buffers(ep)
        dp.reinit_buffers(p)
        dp.update(init, path, env)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[5:])
    return {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "plans": int(len(lat)),
            "what": "reinit_buffers + DynProgLonPlanner.update on the reference default grid 201x37x20, 10 stages, "
                    "host path in, host trajectory out"}


def dp_target_plan(device=0, plans=20):
    """`DynProgLonPlanner.update` on the headline grid (256 x 128 x 64, 50 stages), one plan at a time through the drop-in
    surface: the call takes the chained sweep by default (all stages of the plan in one launch); the same call with
    `set_chained_sweep(0)` beside it."""
    import tpl_amd
    from . import scenarios as sc
    S, V, A, T = sc.GRIDS["target"]
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    env = tpl_amd.DynProgEnvironment(device)
    env.reinit_buffers(ep)
    env.set_occ_map(sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC))
    dp = tpl_amd.DynProgLonPlanner(device)
    dp.reinit_buffers(p)
    init = sc.initial_state(0.0, 8.0, 0.0)
    out = {}
    for rnd in range(2):                                # (the two forms in turn, twice: the better median of each)
        for name, mode in (("chained", -1), ("stage_by_stage", 0)):
            dp.set_chained_sweep(mode)
            lat = []
            for k in range(plans):
                t0 = time.perf_counter()
                dp.update(init, path, env)
                lat.append((time.perf_counter() - t0) * 1e3)
            out[name] = min(out.get(name, 1e9), float(np.percentile(np.array(lat[4:]), 50)))
            out[name + "_ran"] = bool(dp.info()["chained_sweep"])
    return {"p50_ms": out["chained"], "p50_ms_stage_by_stage": out["stage_by_stage"], "chained_sweep": out["chained_ran"],
            "plans": 2 * (plans - 4),
            "what": "DynProgLonPlanner.update on 256x128x64, 50 stages, host path in, host trajectory out; the default (all "
                    "stages of the plan in one chained launch) and set_chained_sweep(0)"}


def env_update(device=0, updates=60, n_objects=10):
    """The occupancy pipeline as dp_env.py:128-170 drives it per planning cycle:
    reinit_buffers + set_ref_line + insert_geometry per predicted object + update()."""
    import tpl_amd
    from . import scenarios as sc
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = sc.env_scene("arc", seed=1, n_objects=n_objects)
    env = tpl_amd.DynProgEnvironment(device)
    lat = []
    for k in range(updates):
        t0 = time.perf_counter()
        env.reinit_buffers(ep)
        env.set_ref_line(rl, step)
        for g, st in objs:
            env.insert_geometry(g, st)
        env.update()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[5:])
    n_tri = sum(len(g) for g, _ in objs) // 3
    return {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "updates": int(len(lat)),
            "triangles": int(n_tri),
            "what": f"reinit_buffers + set_ref_line + {n_objects} x insert_geometry + update(): {n_tri} triangles into an "
                    "840x840 image, 201x31x10 Frenet cells, distance maps; host geometry in"}


def poly_lat_scene(device=0):
    import tpl_amd
    from . import scenarios as sc
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = sc.env_scene("arc", seed=1, n_objects=10)
    env = tpl_amd.DynProgEnvironment(device)
    env.reinit_buffers(ep)
    env.set_ref_line(rl, step)
    for g, st in objs:
        env.insert_geometry(g, st)
    env.update()
    start = tpl_amd.PolyLatTrajPoint()
    start.s, start.l, start.dl, start.v = 5.0, 0.4, 0.05, 8.0
    return ep, rl, step, objs, env, start


def poly_lat_update(device=0, updates=60):
    """PolyLatPlanner.update with the reference's default lattice (21 x 13 quintics x 201 stations)."""
    import tpl_amd
    ep, rl, step, objs, env, start = poly_lat_scene(device)
    p = tpl_amd.PolyLatPlannerParams()
    pl = tpl_amd.PolyLatPlanner(device)
    lat = []
    for k in range(updates):
        start.l = 0.4 + 0.001 * (k % 7)
        t0 = time.perf_counter()
        pl.reinit_buffers(p)
        pl.update(start, env)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[5:])
    return {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "updates": int(len(lat)),
            "what": "reinit_buffers + PolyLatPlanner.update: 21 x 13 quintic paths x 201 stations (local costs, travel "
                    "times, collision look-ups, selection) + final trajectory; host start state in, host trajectory out"}


def lat_lon_update(device=0, updates=40):
    """DynProgLatLonPlanner.update_traj_dp on the reference's default grid (201 x 37 x 21, 10 stages,
    49 actions backward, 441 forward) + smoothing + Cartesian conversion."""
    import tpl_amd
    ep, rl, step, objs, env, _ = poly_lat_scene(device)
    p = tpl_amd.DynProgLatLonPlannerParams()
    p.length_veh, p.width_veh = 4.9, 1.9
    pl = tpl_amd.DynProgLatLonPlanner(device)
    start = tpl_amd.LatLonState()
    start.s, start.ds, start.l = 2.0, 8.0, 0.3
    lat, lat_dp = [], []
    for k in range(updates):
        start.l = 0.3 + 0.001 * (k % 5)
        t0 = time.perf_counter()
        pl.reinit_buffers(p)
        pl.traj_dp[0] = start
        pl.traj_smooth[0] = start
        pl.update_traj_dp(env)
        t1 = time.perf_counter()
        pl.update_traj_smooth(env)
        pl.update_traj_cart(env)
        lat.append((time.perf_counter() - t0) * 1e3)
        lat_dp.append((t1 - t0) * 1e3)
    lat, lat_dp = np.array(lat[5:]), np.array(lat_dp[5:])
    cells = p.s_steps * p.ds_steps * p.l_steps
    return {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
            "update_traj_dp_p50_ms": float(np.percentile(lat_dp, 50)), "updates": int(len(lat)),
            "cell_updates_per_s": float(cells * (p.t_steps - 1) / (np.percentile(lat_dp, 50) * 1e-3)),
            "what": "reinit_buffers + update_traj_dp (9 backward stages of 201x37x21 cells x 49 actions, forward "
                    "rollout of 441 actions per stage) + update_traj_smooth + update_traj_cart; host states in and out"}


def planner_cycles(device=0, cycles=80):
    """Whole planner-plugin cycles (what tplsim times as a planner's update): PolyLatDpLonPlanner.update and
    DpLatLonPlanner.update on the synthetic drive of benchkit/drive.py -- environment update with prediction geometry
    for three objects, previous-plan shift, replan decision, lattice + (s,v,a) plan or (s,ds,l) plan + smoothing,
    trajectory assembly.  Host Python around the engine calls is inside the time, as it is in the reference."""
    from contextlib import contextmanager
    from . import drive
    from tpl_amd.planning.dyn_prog import DpLatLonPlanner, PolyLatDpLonPlanner
    from tpl_amd.store import Bundle

    @contextmanager
    def lock():
        yield
    out = {}
    for name, cls in (("poly_lat_dp_lon_planner", PolyLatDpLonPlanner), ("dp_lat_lon_planner", DpLatLonPlanner)):
        pl = cls(Bundle(), lock)
        d = drive.Drive(v_ego=9.0)
        lat, engine = [], []
        for k in range(cycles):
            sh = d.cycle(k)
            t0 = time.perf_counter()
            tr = pl.update(sh)
            lat.append((time.perf_counter() - t0) * 1e3)
            engine.append(pl.dp_env.runtime_environment)
        lat = np.array(lat[5:])
        out[name] = {"p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "cycles": int(len(lat)),
                     "env_update_p50_ms": float(np.percentile(engine[5:], 50)), "emergency": bool(tr.emergency),
                     "what": f"{cls.__name__}.update on the drive of benchkit/drive.py (10 Hz, 230 m local map, 3 predicted "
                             "objects), default parameters of the plugin, HIP engines; Python glue included"}
    return out


def run(device=0):
    res = {"mpc_closed_loop": mpc_closed_loop(device), "dp_default_plan": dp_default_plan(device), "dp_target_plan": dp_target_plan(device),
           "env_update": env_update(device), "poly_lat_update": poly_lat_update(device),
           "lat_lon_update": lat_lon_update(device)}
    res.update(planner_cycles(device))
    return res
