#!/usr/bin/env python3
"""Runs planning cycles through a module with tplcpp's Python surface and writes what every engine returned.

    python dump_cycle_goldens.py --module tplcpp  --scenes cycle_scenes.npz --out <dir>     # on a CUDA box, in tpl
    python tools/dump_cycle_goldens.py --module tpl_amd                                     # here, on an MI355X

Stand-alone on purpose: numpy + the module under test + the scenes file (inputs only, written by
tools/make_cycle_scenes.py).  The calls are the ones tpl's own planners make: DpEnv.update
(planning/dyn_prog/dp_env.py:130-170), PolyLatDpLonPlanner.update_planner (poly_lat_dp_lon_planner.py:134-181),
DpLatLonPlanner.update_planner (dp_lat_lon_planner.py:136-148).  One ``ref_cycle_<scene>.npz`` per scene;
dropped into tests/golden/ they pin the occupancy pipeline, the lateral lattice, the (s,v,a) planner and the
(s,ds,l) planner against the binary that wrote them (tests/test_cycle_reference_golden.py)."""
import argparse
import importlib
import os

import numpy as np


def fill(obj, names, vals):
    for n, v in zip(names, vals):
        n = str(n)
        cur = getattr(obj, n)
        setattr(obj, n, int(round(float(v))) if isinstance(cur, int) and not isinstance(cur, bool) else float(v))
    return obj


def run_scene(m, z, name):
    g = lambda k: z[f"{name}.{k}"]
    out = {"module": np.array(m.__name__)}

    # ---- occupancy pipeline (dp_env.py:130-170)
    ep = fill(m.DynProgEnvParams(), g("env_names"), g("env_vals"))
    env = m.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(np.ascontiguousarray(g("ref_line"), dtype=np.float64), float(g("ref_step")))
    for i, stationary in enumerate(g("geom_stationary")):
        env.insert_geometry(np.ascontiguousarray(g(f"geom_{i}"), dtype=np.float64), bool(stationary))
    env.update()
    out["occ_cartesian"] = np.asarray(env.get_occ_map_cartesian(), dtype=np.uint32)
    out["occ"] = np.asarray(env.get_occ_map(), dtype=np.float32)
    out["dist_lon"] = np.asarray(env.get_dist_map_lon(), dtype=np.float32)

    # ---- lateral lattice (poly_lat_dp_lon_planner.py:138-142)
    lp = fill(m.PolyLatPlannerParams(), g("lat_names"), g("lat_vals"))
    lat = m.PolyLatPlanner()
    lat.reinit_buffers(lp)
    start = m.PolyLatTrajPoint()
    start.s, start.l, start.dl, start.ddl, start.v = (float(v) for v in g("lat_start"))
    traj_lat = lat.update(start, env)
    out["lat_traj"] = np.asarray(traj_lat.as_numpy(), dtype=np.float64)

    # ---- path for the speed planner (poly_lat_dp_lon_planner.py:150-155; the v_max shaping of :157-168 is host
    #      Python on both sides and left out) and the (s,v,a) plan (:176-181)
    p = fill(m.DynProgLonPlannerParams(), g("lon_names"), g("lon_vals"))
    samples = np.asarray(traj_lat.lerp(np.arange(0.0, p.path_steps * p.path_step_size, p.path_step_size)))
    path = np.ascontiguousarray(samples[:, [7, 8, 5, 1, 11, 6, 10]], dtype=np.float32)
    out["lon_path"] = path
    dp = m.DynProgLonPlanner()
    dp.reinit_buffers(p)
    st = m.LonState()
    st.s, st.v, st.a = (float(v) for v in g("lon_start"))
    traj_lon = dp.update(st, path, env)
    out["lon_traj"] = np.array([[s.t, s.s, s.v, s.a, s.j, s.cost, s.constr] for s in traj_lon.states], dtype=np.float32)
    if hasattr(dp, "copy_table"):                       # not in tplcpp's bindings; see INTEGRATION.md for the C++ side
        for t in (1, p.t_steps - 1):
            out[f"lon_table_{t}"] = np.asarray(dp.copy_table(t), dtype=np.float32)

    # ---- (s, ds, l) planner (dp_lat_lon_planner.py:136-148)
    q = fill(m.DynProgLatLonPlannerParams(), g("ll_names"), g("ll_vals"))
    ll = m.DynProgLatLonPlanner()
    ll.reinit_buffers(q)
    s0 = m.LatLonState()
    s0.s, s0.ds, s0.l = (float(v) for v in g("ll_start"))
    ll.traj_dp[0] = s0
    ll.traj_smooth[0] = s0
    ll.update_traj_dp(env)
    out["ll_traj_dp"] = np.asarray(ll.traj_dp.as_numpy(), dtype=np.float64)
    for t in (1, q.t_steps - 1):
        out[f"ll_table_{t}"] = np.asarray(ll.copy_backward_pass_tex(t), dtype=np.float32)
    ll.update_traj_smooth(env)
    ll.update_traj_cart(env.ref_line)
    out["ll_traj_smooth"] = np.asarray(ll.traj_smooth.as_numpy(), dtype=np.float64)
    out["ll_traj_smooth_cart"] = np.asarray(ll.traj_smooth_cart.as_numpy(), dtype=np.float64)
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument("--module", default="tplcpp")
    ap.add_argument("--scenes", default=os.path.join(here, "..", "tests", "golden", "cycle_scenes.npz"))
    ap.add_argument("--out", default=".")
    ap.add_argument("--only", default=None, help="one scene name")
    args = ap.parse_args()
    m = importlib.import_module(args.module)
    z = np.load(args.scenes)
    os.makedirs(args.out, exist_ok=True)
    for name in z["scenes"]:
        name = str(name)
        if args.only and name != args.only:
            continue
        res = run_scene(m, z, name)
        res["scene"] = np.array(name)
        path = os.path.join(args.out, f"ref_cycle_{name}.npz")
        np.savez_compressed(path, **res)
        print(path)


if __name__ == "__main__":
    main()
