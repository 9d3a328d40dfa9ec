#!/usr/bin/env python3
"""Writes tests/golden/cycle_scenes.npz: the INPUTS of three planning cycles (reference line, prediction
triangles, every parameter object by attribute name, start states).  tools/dump_cycle_goldens.py reads this
file and needs nothing else from this repository, so that a tpl maintainer can run it against ``tplcpp``.
Seeded; re-running reproduces the committed file."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpl_amd                                      # noqa: E402
from benchkit import scenarios as sc                 # noqa: E402


def named(p):
    names = list(p.__slots__)
    return np.array(names), np.array([float(getattr(p, n)) for n in names], dtype=np.float64)


def scene(kind, seed, n_objects, lon_grid, ll_grid):
    out = {}
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = sc.env_scene(kind, seed=seed, n_objects=n_objects)
    out["env_names"], out["env_vals"] = named(ep)
    out["ref_line"], out["ref_step"] = rl, np.float64(step)
    out["geom_stationary"] = np.array([st for _, st in objs], dtype=bool)
    for i, (g, _) in enumerate(objs):
        out[f"geom_{i}"] = g
    lp = tpl_amd.PolyLatPlannerParams()
    out["lat_names"], out["lat_vals"] = named(lp)
    out["lat_start"] = np.array([2.0, 0.3, 0.02, 0.0, 8.0])            # s, l, dl, ddl, v
    p = sc.lon_params(*lon_grid)
    out["lon_names"], out["lon_vals"] = named(p)
    out["lon_start"] = np.array([0.0, 8.0, 0.0])                       # s, v, a
    q = tpl_amd.DynProgLatLonPlannerParams()
    q.s_steps, q.ds_steps, q.l_steps, q.t_steps = ll_grid
    q.ds_max = 18.0
    q.length_veh, q.width_veh = 4.9, 1.9
    out["ll_names"], out["ll_vals"] = named(q)
    out["ll_start"] = np.array([2.0, 7.0, 0.3])                        # s, ds, l
    return out


def main():
    scenes = {
        "straight": scene("straight", 0, 6, (201, 37, 20, 10), (101, 19, 11, 7)),
        "arc": scene("arc", 1, 8, (201, 37, 20, 10), (101, 19, 11, 7)),
        "arc_dense": scene("arc", 7, 12, (128, 64, 32, 10), (151, 25, 15, 9)),
    }
    flat = {"scenes": np.array(sorted(scenes))}
    for name, d in scenes.items():
        for k, v in d.items():
            flat[f"{name}.{k}"] = v
    path = os.path.join(ROOT, "tests", "golden", "cycle_scenes.npz")
    np.savez_compressed(path, **flat)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
