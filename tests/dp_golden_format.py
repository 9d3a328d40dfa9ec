"""Dump format that pins HP-1 against the reference binary, and its loader.

The reference's DP (s,v,a) planner needs CUDA and cannot run in this image, so the repository
holds no reference vector for it ("parity unpinned").  A tpl maintainer with a CUDA box
closes that gap with ONE file per scenario, ``tests/golden/dp_lon_<name>.npz``, written by the
20-line recipe in INTEGRATION.md ("Pinning HP-1").  Its arrays:

  lon_params   float64[28]  DynProgLonPlannerParams in the field order of lon_planner.cuh:7-53
               (= include/tplhip.h TplDpLonParams; the three derived step sizes may be 0)
  env_params   float64[19]  DynProgEnvParams in the field order of env.cuh:11-44 (TplDpEnvParams)
  path         float32[path_steps, 7]   the array handed to DynProgLonPlanner.update
  occ          float32[l_steps, s_steps, t_steps]   env.get_occ_map() right before the update
  inits        float32[n, 7]            LonState rows (t, s, v, a, j, cost, constr) handed to update
  trajs        float32[n, t_steps, 7]   LonTraj.states of the result, same row layout
  table_<t>    float32[a_steps, v_steps, s_steps, 4]   OPTIONAL: stage t of the value table as
               surf3Dwrite stores it (needs a few lines of C++ on the CUDA side)

What is compared (tests/test_dp_reference_golden.py), for the CPU restatement and for the HIP planner:
chosen ACTION INDICES of every trajectory node bit-exact, states and costs within 1e-5 relative,
and -- where tables were dumped -- argmin actions per cell and values within 1e-5 relative.
Test infrastructure; nothing under tpl_amd/ imports this module.
"""
import ctypes as C
import glob
import os

import numpy as np

from tpl_amd._capi import TplDpEnvParams, TplDpLonParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_GLOB = os.path.join(ROOT, "tests", "golden", "dp_lon_*.npz")
REL_TOL = 1e-5


def golden_files():
    return sorted(glob.glob(GOLDEN_GLOB))


def struct_to_vec(c):
    return np.array([float(getattr(c, n)) for n, _ in c._fields_], dtype=np.float64)


def vec_to_struct(vec, ctype):
    c = ctype()
    fields = ctype._fields_
    if len(vec) != len(fields):
        raise ValueError(f"{ctype.__name__}: expected {len(fields)} values, got {len(vec)}")
    for (name, ft), v in zip(fields, vec):
        setattr(c, name, int(round(float(v))) if ft is C.c_int32 else float(v))
    return c


def save(path, lon_params_c, env_params_c, path_arr, occ, inits, trajs, tables=None):
    """Writer of the format (used by the loader's own round-trip test; the CUDA-side recipe writes
    the same arrays with plain numpy)."""
    arrs = dict(lon_params=struct_to_vec(lon_params_c), env_params=struct_to_vec(env_params_c),
                path=np.asarray(path_arr, np.float32), occ=np.asarray(occ, np.float32),
                inits=np.asarray(inits, np.float32), trajs=np.asarray(trajs, np.float32))
    for t, tab in (tables or {}).items():
        arrs[f"table_{int(t)}"] = np.asarray(tab, np.float32)
    np.savez_compressed(path, **arrs)


def load(path):
    z = np.load(path)
    need = ("lon_params", "env_params", "path", "occ", "inits", "trajs")
    missing = [k for k in need if k not in z.files]
    if missing:
        raise ValueError(f"{os.path.basename(path)}: missing arrays {missing}")
    p = vec_to_struct(z["lon_params"], TplDpLonParams)
    ep = vec_to_struct(z["env_params"], TplDpEnvParams)
    g = dict(p=p, ep=ep, path=np.ascontiguousarray(z["path"], np.float32), occ=np.ascontiguousarray(z["occ"], np.float32),
             inits=np.asarray(z["inits"], np.float32), trajs=np.asarray(z["trajs"], np.float32),
             tables={int(k[6:]): np.asarray(z[k], np.float32) for k in z.files if k.startswith("table_")})
    if g["path"].shape != (p.path_steps, 7):
        raise ValueError(f"path has shape {g['path'].shape}, parameters say {(p.path_steps, 7)}")
    if g["occ"].shape != (ep.l_steps, ep.s_steps, ep.t_steps):
        raise ValueError(f"occ has shape {g['occ'].shape}, parameters say {(ep.l_steps, ep.s_steps, ep.t_steps)}")
    n = g["inits"].shape[0]
    if g["inits"].shape != (n, 7) or g["trajs"].shape != (n, p.t_steps, 7):
        raise ValueError(f"inits {g['inits'].shape} / trajs {g['trajs'].shape} do not match t_steps = {p.t_steps}")
    for t, tab in g["tables"].items():
        if tab.shape != (p.a_steps, p.v_steps, p.s_steps, 4) or not (1 <= t < p.t_steps):
            raise ValueError(f"table_{t} has shape {tab.shape}")
    return g


def action_index(j, p, n):
    """Index of a jerk among the n candidates j_min + (j_max - j_min) k/(n-1) (lon_planner.cu:148-149)."""
    span = float(p.j_max) - float(p.j_min)
    return np.rint((np.asarray(j, np.float64) - float(p.j_min)) / span * (n - 1)).astype(np.int64)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-12))) if a.size else 0.0


def score_against(g, trajs, tables=None):
    """How far a plan is from the dump, as numbers instead of a verdict: trajectory nodes / table cells with another
    action and the largest relative value differences (what tests/test_dp_reference_golden.py tabulates per variant)."""
    p = g["p"]
    want = g["trajs"]
    ka, kb = action_index(trajs[..., 4], p, 21), action_index(want[..., 4], p, 21)
    out = {"traj_nodes": int(ka[:, :-1].size), "traj_nodes_other_action": int(np.count_nonzero(ka[:, :-1] != kb[:, :-1])),
           "traj_value_rel": max(rel_err(trajs[..., c], want[..., c]) for c in (1, 2, 3, 5, 6)),
           "table_cells": 0, "table_cells_other_action": 0, "table_value_rel": 0.0}
    for t, ref in g["tables"].items():
        got = tables[t]
        out["table_cells"] += int(ref[..., 2].size)
        out["table_cells_other_action"] += int(np.count_nonzero(action_index(got[..., 2], p, 9) != action_index(ref[..., 2], p, 9)))
        out["table_value_rel"] = max(out["table_value_rel"], rel_err(got[..., 0], ref[..., 0]), rel_err(got[..., 1], ref[..., 1]))
    return out


def check_against(g, trajs, tables=None, who=""):
    """trajs: float32[n, T, 7] computed on g's inputs; tables: {t: float32[a, v, s, 4]} or None.
    Raises AssertionError naming the first thing outside the bar."""
    p = g["p"]
    want = g["trajs"]
    assert trajs.shape == want.shape, (who, trajs.shape, want.shape)
    ka, kb = action_index(trajs[..., 4], p, 21), action_index(want[..., 4], p, 21)
    # the last node's jerk is whatever the caller passed in (terminal stage chooses none)
    bad = np.argwhere(ka[:, :-1] != kb[:, :-1])
    assert bad.size == 0, f"{who}: {len(bad)} trajectory nodes chose another action, first (rollout, stage) = {tuple(bad[0])}"
    for col, name in ((1, "s"), (2, "v"), (3, "a"), (5, "cost"), (6, "constr")):
        e = rel_err(trajs[..., col], want[..., col])
        assert e <= REL_TOL, f"{who}: trajectory {name} differs by {e:.2e} relative (> {REL_TOL})"
    for t, ref in g["tables"].items():
        assert tables is not None and t in tables, f"{who}: stage {t} was dumped but not computed"
        got = tables[t]
        flips = np.count_nonzero(action_index(got[..., 2], p, 9) != action_index(ref[..., 2], p, 9))
        assert flips == 0, f"{who}: stage {t}: {flips} cells chose another action"
        for col, name in ((0, "cost"), (1, "constr")):
            e = rel_err(got[..., col], ref[..., col])
            assert e <= REL_TOL, f"{who}: stage {t} {name} differs by {e:.2e} relative"
