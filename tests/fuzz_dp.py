#!/usr/bin/env python3
"""Randomised parity soak of the DP (s,v,a) planner: N random configurations (grid sizes, ranges, weights, stage lengths,
path step sizes that are and are not powers of two, speed-limit profiles, obstacle sets, start states), every table
and trajectory of the HIP planner compared bit for bit with oracle/dp_lon_oracle.c.  Not collected by pytest (a soak
tool; tests/test_dp_gpu.py::test_fuzz_sample runs a short sample of it).

    python tests/fuzz_dp.py [cases] [first seed] [--rules]

--rules: every case also draws one of the 18 arithmetic rule combinations (R2 weight rule x R3 blend form x R1
contraction: set_filter_rule / set_contraction) and is compared with the oracle build that makes the same choices
(the ONE checker library under dpo_set_rule(quant, lerp, fma, order)).
--orders: with --rules, every case also draws R3's blend order (set_blend_order: 0 a->v->s, 1 s->v->a, 2 / 3 the
eight-corner sums) -- 72 combinations."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_dp as O                       # noqa: E402
import tpl_amd                              # noqa: E402
from benchkit import scenarios as sc         # noqa: E402


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.choice([33, 64, 96, 128, 201, 256, 300]))
    V = int(rng.integers(2, 40))
    A = int(rng.integers(2, 24))
    T = int(rng.integers(2, 7))
    dt = float(rng.choice([0.5, 0.8, 1.0, 1.25]))
    p = sc.lon_params(S, V, A, T, dt=dt, dt_start=float(rng.choice([dt, 0.3 * dt, 0.71 * dt])))
    p.s_max = float(rng.choice([100.0, 200.0, 157.3]))
    p.v_max = float(rng.choice([20.0, 36.0, 27.5]))
    p.a_min, p.a_max = float(rng.uniform(-4.0, -1.0)), float(rng.uniform(0.5, 2.5))
    p.j_min, p.j_max = float(rng.uniform(-4.0, -1.0)), float(rng.uniform(0.5, 3.0))
    p.time_gap, p.gap_min = float(rng.uniform(0.0, 2.0)), float(rng.uniform(0.0, 3.0))
    for w in ("w_progress", "w_a", "w_j", "w_snap", "w_safety_dist"):
        setattr(p, w, float(rng.choice([0.0, 0.1, 0.5, 1.0, 3.7, 10.0])))
    p.path_step_size = float(rng.choice([0.5, 1.0, 2.0, 0.7, 1.3]))             # powers of two take the cheap division
    p.path_steps = int(np.ceil(p.s_max / p.path_step_size)) + int(rng.integers(0, 20))
    ep = sc.env_params(t_steps=T, s_steps=int(rng.choice([101, 201])), l_steps=int(rng.choice([11, 31])), dt=dt,
                       dt_start=p.dt_start)
    ep.s_max = p.s_max
    d = np.arange(p.path_steps, dtype=np.float64) * p.path_step_size
    path = np.zeros((p.path_steps, 7))
    path[:, 0] = d
    path[:, 2] = d * float(rng.choice([1.0, 0.97]))                              # stations lag the distance on a curve
    path[:, 3] = rng.uniform(-1.0, 1.0)
    path[:, 5] = np.clip(rng.uniform(6.0, 16.0) - rng.uniform(0.0, 8.0) * np.exp(-((d - rng.uniform(30, 150)) / rng.uniform(5, 30)) ** 2)
                         + rng.normal(0.0, 0.2, len(d)), 0.0, None)
    if rng.random() < 0.3:
        path[int(rng.integers(10, len(d))):, 5] = 0.0                             # a stop point
    path[:, 6] = d
    obstacles = sc.random_obstacles(int(rng.integers(0, 1000)), int(rng.integers(0, 9)))
    occ = sc.occ_map(ep, obstacles)
    inits = [sc.initial_state(float(rng.uniform(0.0, 30.0)), float(rng.uniform(0.0, 15.0)), float(rng.uniform(-1.5, 1.0)))
             for _ in range(3)]
    return p, ep, path.astype(np.float32), occ, inits


def run_case(seed, rules=False, orders=False):
    p, ep, path, occ, inits = random_case(seed)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    rule = None
    if rules:
        rng = np.random.default_rng(77 + seed)
        rule = (int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 4)) if orders else 0)
        dp.set_contraction(bool(rule[2]))
        dp.set_filter_rule(rule[0], rule[1])
        dp.set_blend_order(rule[3])
        O.set_rule(quant=rule[0], lerp=rule[1], fma=rule[2], order=rule[3])
    trajs = dp.update_batch(inits, path, env)
    try:
        tables, otrajs, dist = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
    finally:
        if rules:
            O.set_rule()
    assert np.array_equal(env.get_dist_map_path().view(np.uint32), dist.view(np.uint32)), f"seed {seed}: distance map"
    for t in range(1, p.t_steps):
        got = dp.copy_table(t)
        bad = np.count_nonzero(got.view(np.uint32) != tables[t].view(np.uint32))
        assert bad == 0, f"seed {seed}: stage {t}: {bad} of {got.size} words differ ({dp.info()})"
    for r, tr in enumerate(trajs):
        got = np.array([[s.t, s.s, s.v, s.a, s.j, s.cost, s.constr] for s in tr.states], dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), otrajs[r].view(np.uint32)), f"seed {seed}: trajectory {r}"
    info = dp.info()
    return info["sweep"] + (" (chained)" if info.get("chained_launches", 0) else "")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rules = "--rules" in sys.argv
    orders = "--orders" in sys.argv
    n = int(args[0]) if len(args) > 0 else 100
    first = int(args[1]) if len(args) > 1 else 0
    kinds = {}
    for seed in range(first, first + n):
        k = run_case(seed, rules, orders)
        kinds[k] = kinds.get(k, 0) + 1
    print(f"{n} random configurations{' under random arithmetic rules' if rules else ''}{' and blend orders' if rules and orders else ''} bit-identical; sweep kernels taken: {kinds}")


if __name__ == "__main__":
    main()
