#!/usr/bin/env python3
"""Randomised soak of the scenario-batched DP launches: random grids, scenario counts, chunk sizes, per-scenario paths or
one shared path, random obstacle sets and start states -- every scenario's trajectory bit-identical to the CPU
restatement's plan of that scenario, spot stages of its value table too, and the device-side best-plan record equal to
the host-side reduction of the CPU plans.

    python tests/fuzz_scenarios.py [cases] [first seed]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_dp as O                                   # noqa: E402
import tpl_amd                                          # noqa: E402
from benchkit import scenarios as sc
from tpl_amd import sharding           # noqa: E402


def run_case(seed):
    rng = np.random.default_rng(9000 + seed)
    S, V, A, T = int(rng.choice([64, 96, 128, 201])), int(rng.integers(4, 40)), int(rng.integers(3, 20)), int(rng.integers(3, 7))
    n, chunk = int(rng.integers(1, 12)), int(rng.integers(1, 6))
    own_paths = bool(rng.random() < 0.5)
    p = sc.lon_params(S, V, A, T)
    if rng.random() < 0.25:
        p.path_step_size, p.path_steps = 0.7, 290                 # no cheap division: the general kernel
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=int(rng.choice([11, 31])))
    occs = np.stack([sc.occ_map(ep, sc.random_obstacles(int(rng.integers(0, 10**6)), int(rng.integers(0, 7)))) for _ in range(n)])
    inits = [sc.initial_state(float(rng.uniform(0, 20)), float(rng.uniform(1.0, 14.0)), float(rng.uniform(-1.0, 1.0))) for _ in range(n)]
    base = sc.straight_path(p.path_steps, p.path_step_size)
    if own_paths:
        paths = np.stack([base.copy() for _ in range(n)])
        for i in range(n):
            paths[i, :, 5] = np.clip(paths[i, :, 5] + rng.uniform(-4.0, 3.0), 0.0, None)
            paths[i, :, 3] = rng.uniform(-1.0, 1.0)
    else:
        paths = base[None]
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    dp.upload_scenarios(occs, ep, inits, paths, chunk=chunk)
    cp, cep = O.cparams(p), O.cparams(ep)
    want, records = [], []
    for first in range(0, n, chunk):
        count = min(chunk, n - first)
        dp.plan_scenarios(first, count, id_base=500, reset_best=(first == 0))
        for i in range(first, first + count):
            tables, otrajs, _ = O.update(cp, cep, paths[i if own_paths else 0], occs[i], [inits[i]])
            t = int(rng.integers(1, T))
            got = dp.copy_scenario_table(i, t)
            assert np.array_equal(got.view(np.uint32), tables[t].view(np.uint32)), f"seed {seed}: scenario {i} stage {t}"
            want.append(otrajs[0])
            records.append(sharding.plan_record(otrajs[0], 500 + i))
    got = dp.download_scenarios()
    for i in range(n):
        g = O.states_to_np([s._c for s in got[i].states])
        assert np.array_equal(g.view(np.uint32), want[i].view(np.uint32)), f"seed {seed}: trajectory {i}"
    best = sharding.local_best(records)
    constr, cost, rid, states = dp.best_record()
    assert (constr, cost, rid) == (np.float32(best[0]), np.float32(best[1]), int(best[2])), f"seed {seed}: best record"
    return dp.info()["scenario_kernel"], n


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kinds, total = {}, 0
    for seed in range(first, first + cases):
        k, n = run_case(seed)
        kinds[k] = kinds.get(k, 0) + 1
        total += n
    print(f"{cases} random batches ({total} scenarios) bit-identical, best-plan records equal; scenario kernels taken: {kinds}")


if __name__ == "__main__":
    main()
