#!/usr/bin/env python3
"""Soak of the chained sweep (SweepChain): planner A sweeps stage by stage, planner B in one chained launch per plan, same
inputs -- a new obstacle set, start states and path speed limit per plan.  Compared as raw bits: every trajectory and
`tables` randomly chosen stage tables per plan (all of them on the small grids).  A stale row in some XCD's L2, a flag seen
before its rows, a dependency missing from a workgroup's list would all show here as a differing table word.
usage: python3 tools/chain_soak.py [plans on 256x128x64] [plans per small grid] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpl_amd  # noqa: E402
from benchkit import scenarios as sc  # noqa: E402


def random_inputs(p, ep, rng):
    path = sc.straight_path(p.path_steps, p.path_step_size)
    path[:, 5] = rng.uniform(6.0, 16.0) - rng.uniform(0.0, 6.0) * np.exp(-((path[:, 6] - rng.uniform(40, 160)) / rng.uniform(8, 30)) ** 2)
    occ = sc.occ_map(ep, sc.random_obstacles(int(rng.integers(1 << 30)), int(rng.integers(0, 9))))
    inits = sc.random_initial_states(int(rng.integers(1, 5)), seed=int(rng.integers(1 << 30)))
    return path, occ, inits


def soak_side_by_side(S, V, A, T, plans, rng):
    """TWO chained planners busy at once (plans enqueued on both before either is read), each against a stage-by-stage one."""
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    ref = tpl_amd.DynProgLonPlanner(); ref.reinit_buffers(p); ref.set_chained_sweep(0)
    renv = tpl_amd.DynProgEnvironment(); renv.reinit_buffers(ep)
    lanes = []
    for _ in range(2):
        env = tpl_amd.DynProgEnvironment(); env.reinit_buffers(ep)
        dp = tpl_amd.DynProgLonPlanner(); dp.reinit_buffers(p); dp.set_chained_sweep(1)
        lanes.append((dp, env))
    for k in range(plans):
        ins = [random_inputs(p, ep, rng) for _ in lanes]
        for (dp, env), (path, occ, inits) in zip(lanes, ins):
            env.set_occ_map(occ); dp.upload(inits, path); dp.plan_resident(env)
        for li, ((dp, env), (path, occ, inits)) in enumerate(zip(lanes, ins)):
            got = dp.download()
            renv.set_occ_map(occ)
            want = ref.update_batch(inits, path, renv)
            for r, (a, b) in enumerate(zip(want, got)):
                assert np.array_equal(a.as_numpy().view(np.uint64), b.as_numpy().view(np.uint64)), (k, li, r)
            t = int(rng.integers(1, T - 1))
            x, y = ref.copy_table(t), dp.copy_table(t)
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (k, li, t)
    assert all(dp.info()["chained_launches"] == plans for dp, _ in lanes)


def soak(S, V, A, T, plans, tables, rng, resident):
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    envs, dps = [], []
    rule = (1, 0, bool(rng.integers(0, 2)))            # (the chained form is built for the default filter rule under either R1)
    for mode in (0, 1):
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p)
        dp.set_filter_rule(rule[0], rule[1])
        dp.set_contraction(rule[2])
        dp.set_chained_sweep(mode)
        envs.append(env); dps.append(dp)
    words = 0
    for k in range(plans):
        path, occ, inits = random_inputs(p, ep, rng)
        out = []
        for env, dp in zip(envs, dps):
            env.set_occ_map(occ)
            if resident:
                dp.upload(inits, path); dp.plan_resident(env); out.append(dp.download())
            else:
                out.append(dp.update_batch(inits, path, env))
        for r, (a, b) in enumerate(zip(*out)):
            x, y = a.as_numpy(), b.as_numpy()
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), (S, V, A, T, k, r)
        stages = range(1, T) if tables >= T - 1 else rng.choice(np.arange(1, T - 1), size=tables, replace=False)
        for t in stages:
            x, y = dps[0].copy_table(int(t)), dps[1].copy_table(int(t))
            if not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
                bad = np.argwhere(x.view(np.uint32) != y.view(np.uint32))
                raise AssertionError(f"{S}x{V}x{A}x{T} plan {k} stage {t}: {len(bad)} words differ, first at {tuple(bad[0])}")
            words += x.size
    i0, i1 = dps[0].info(), dps[1].info()
    assert i1["chained_launches"] == plans and i0["chained_launches"] == 0, (i0, i1)
    return words, rule


def main():
    big = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    small = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    total = 0
    for (S, V, A, T, n, tb, res) in [(256, 128, 64, 50, big, 3, True), (256, 64, 64, 12, small, 99, False), (128, 128, 64, 9, small, 99, True),
                                     (128, 65, 128, 6, small, 99, False), (512, 64, 32, 8, small, 99, True), (512, 128, 32, 10, small // 2, 4, False)]:
        if n <= 0:
            continue
        w, rule = soak(S, V, A, T, n, tb, rng, res)
        total += w
        print(f"{S}x{V}x{A}x{T}: {n} plans chained == stage by stage under rule (quant {rule[0]}, lerp {rule[1]}, R1 {int(rule[2])}) "
              f"({w / 1e6:.0f} M table words compared, {'plan_resident' if res else 'update_batch'}), {time.time() - t0:.0f} s", flush=True)
    if big > 0:
        soak_side_by_side(256, 128, 64, 50, max(4, big // 2), rng)
        print(f"256x128x64x50: {max(4, big // 2)} x 2 plans on TWO chained planners at once == stage by stage, {time.time() - t0:.0f} s", flush=True)
    print(f"chain soak seed {seed}: all equal, {total / 1e9:.2f} G table words")


if __name__ == "__main__":
    main()
