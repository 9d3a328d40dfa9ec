#!/usr/bin/env python3
"""How far do the frozen arithmetic rules of the HP-1 restatement move its results?

The reference's DP (s,v,a) planner is CUDA (nvcc + texture hardware) and cannot run here, so the
oracle freezes choices the reference BINARY may make differently (oracle/dp_lon_oracle.c, rules
R1-R3): which multiply-adds are contracted, texture weights rounded to 8 fractional bits with rint,
blend formula fma(q, w, p(1-w)), blend order a -> v -> s.  This script runs the ONE checker library
under each choice flipped (dpo_set_rule) and measures, against the default rule on identical inputs:

  * the fraction of value-table cells whose ARGMIN JERK differs (stored per cell, all stages),
  * the largest relative difference of the stored cost / constraint values,
  * the largest deviation of the planned trajectories (s, v, a) and the fraction of trajectory
    nodes whose chosen jerk differs,

on BASELINE configs[1] (128x64x32, 50 stages) and on the reference's default grid (201x37x20, 10
stages).  Output: profiles/r06_dp_parity_sensitivity.json and a markdown table on stdout (DESIGN.md
2.4).  Since round 6 the BASE is the contracted default (R1 = 1); rounds 2-5 measured against R1 = 0.

    python tests/dp_sensitivity_study.py [--quick] [--variants=a,b]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_dp as O                      # noqa: E402
from benchkit import scenarios as sc        # noqa: E402

VARIANTS = [
    ("unfused", dict(fma=0), "R1 = 0: one rounding per source operation, nothing contracted (the default of rounds 1-5)"),
    ("wfloor", dict(quant=2), "filter weight truncated to 8 fractional bits (floor) instead of rint"),
    ("wnone", dict(quant=0), "un-quantised filter weights"),
    ("lerp1", dict(lerp=1), "blend p(1-w) + qw, every operation rounded"),
    ("lerp2", dict(lerp=2), "blend p + w(q-p)"),
    ("unfused_wfloor", dict(fma=0, quant=2), "nothing contracted AND truncated weights"),
    # rule R3's ORDER and the symmetric form (round 5)
    ("order_sva", dict(order=1), "blend ORDER s -> v -> a (x first) instead of a -> v -> s, same 1-D lerp"),
    ("sum8_f32", dict(order=2), "symmetric eight-corner sum, exact weight products, fp32 accumulator in the order of the documented formula"),
    ("sum8_wide", dict(order=3), "the same sum with exact products in a double accumulator, one rounding"),
]

CASES = {
    "config2 128x64x32x50": dict(grid="config2", obstacles=sc.OBSTACLES_OVERTAKE),
    "reference default 201x37x20x10": dict(grid="default", obstacles=sc.OBSTACLES_OVERTAKE_TRAFFIC),
}


def plan(case, n_inits):
    S, V, A, T = sc.GRIDS[case["grid"]]
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, case["obstacles"])
    inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(n_inits - 1, seed=5)
    tables, trajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
    return tables, trajs


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)))


def action_index(j, n):
    """Index of the jerk among the n candidates (lon_planner.cu:148-149).  A contracted build rounds
    j_min + (j_max - j_min) * k/(n-1) differently, so jerks are compared as INDICES, not as floats."""
    return np.rint((j.astype(np.float64) + 3.0) / 5.0 * (n - 1)).astype(np.int32)      # sc.lon_params: j in [-3, 2]


def compare(base, var):
    tb, jb = base
    tv, jv = var
    tb, tv = tb[1:], tv[1:]                               # stage 0 is never written
    cells = tb[..., 0].size
    same = action_index(tb[..., 2], 9) == action_index(tv[..., 2], 9)
    flips = int(cells - np.count_nonzero(same))
    traj_nodes = jb[..., 4].size
    return {
        "argmin_jerk_flip_fraction": flips / cells,
        "argmin_jerk_flips": flips, "cells": int(cells),
        # a flipped lexicographic argmin can land on an action of quite different cost (the constraint
        # value decides first), so the value differences are given for the cells that kept their action too
        "max_rel_cost": rel(tb[..., 0], tv[..., 0]),
        "max_rel_cost_same_action": rel(tb[..., 0][same], tv[..., 0][same]),
        "max_rel_constr": rel(tb[..., 1], tv[..., 1]),
        "max_abs_constr": float(np.max(np.abs(tb[..., 1] - tv[..., 1]))),
        "max_abs_constr_same_action": float(np.max(np.abs(tb[..., 1][same] - tv[..., 1][same]))),
        "traj_jerk_flip_fraction": float(np.count_nonzero(action_index(jb[..., 4], 21) != action_index(jv[..., 4], 21))) / traj_nodes,
        "traj_max_abs_s": float(np.max(np.abs(jb[..., 1] - jv[..., 1]))),
        "traj_max_abs_v": float(np.max(np.abs(jb[..., 2] - jv[..., 2]))),
        "traj_max_abs_a": float(np.max(np.abs(jb[..., 3] - jv[..., 3]))),
        "traj_max_rel_cost": rel(jb[..., 5], jv[..., 5]),
        "plans_identical": int(np.count_nonzero(np.all(action_index(jb[..., 4], 21) == action_index(jv[..., 4], 21), axis=1))),
        "plans": int(jb.shape[0]),
    }


def main():
    quick = "--quick" in sys.argv
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--variants=")]
    variants = [(v, kw, d) for v, kw, d in VARIANTS if not only or v in only[0]]
    n_inits = 8 if quick else 64
    out = {"what": __doc__.split("\n\n")[0], "initial_states_per_case": n_inits, "cases": {}}
    for cname, case in CASES.items():
        if quick and "config2" in cname:
            continue
        O.set_rule()
        base = plan(case, n_inits)
        res = {}
        for v, kw, desc in variants:
            with O.rule(**kw):
                res[v] = dict(compare(base, plan(case, n_inits)), variant=desc, rule=O.get_rule() if False else dict(O.DEFAULT_RULE, **kw))
            print(f"[{cname}] {v}: {res[v]}", file=sys.stderr, flush=True)
        out["cases"][cname] = res
    dst = os.path.join(ROOT, "profiles", "r06_dp_parity_sensitivity.json")
    if not quick:
        with open(dst, "w") as fd:
            json.dump(out, fd, indent=1)
    print("| grid | variant | cells with another argmin action | max rel cost, same action (all cells) | max abs constr, same action (all) | trajectory nodes with another action | max traj dev s / v / a | plans with identical actions |")
    print("|---|---|---|---|---|---|---|---|")
    for cname, res in out["cases"].items():
        for v, r in res.items():
            print(f"| {cname} | {v} | {r['argmin_jerk_flip_fraction']:.2e} ({r['argmin_jerk_flips']} of {r['cells']}) | "
                  f"{r['max_rel_cost_same_action']:.1e} ({r['max_rel_cost']:.1e}) | {r['max_abs_constr_same_action']:.1e} ({r['max_abs_constr']:.1e}) | "
                  f"{r['traj_jerk_flip_fraction']:.2e} | {r['traj_max_abs_s']:.1e} m / {r['traj_max_abs_v']:.1e} m/s / "
                  f"{r['traj_max_abs_a']:.1e} m/s2 | {r['plans_identical']} of {r['plans']} |")


if __name__ == "__main__":
    main()
