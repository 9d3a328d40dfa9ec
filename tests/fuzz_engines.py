#!/usr/bin/env python3
"""Randomised parity soak of the three other DP-family engines against their CPU restatements: N random scenes
(straight / arc reference lines with random radius and heading, 0..14 predicted objects, grid sizes, dilation, stage
length) -> occupancy image, Frenet occupancy and distance map bit for bit; the lateral lattice (values 1e-5, discrete
outcomes identical, the final trajectory to 1e-12); the (s,ds,l) tables and trajectory bit for bit.

    python tests/fuzz_engines.py [scenes] [first seed] [--rules]

--rules: the (s,ds,l) planner of every scene also draws one of the 36 texture-filter rules (weight rule x blend form x blend
order / eight-corner sums: DynProgLatLonPlanner.set_filter_rule) and is compared with the build of oracle/lat_lon_oracle.c that
makes the same choice."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_env as oe                         # noqa: E402
import oracle_lat_lon as oll                    # noqa: E402
import oracle_poly_lat as opl                   # noqa: E402
import tpl_amd                                  # noqa: E402
from benchkit import scenarios as sc             # noqa: E402


def run_scene(seed, rules=False):
    rng = np.random.default_rng(5000 + seed)
    T = int(rng.integers(3, 12))
    dt = float(rng.choice([0.5, 1.0, 0.8]))
    ep = sc.env_params(t_steps=T, s_steps=int(rng.choice([101, 128, 201])), l_steps=int(rng.choice([11, 21, 31])), dt=dt)
    ep.s_max = float(rng.choice([120.0, 200.0]))
    ep.l_min, ep.l_max = -float(rng.integers(3, 7)), float(rng.integers(3, 7))
    ep.dilation = float(rng.uniform(0.3, 1.8))
    if rng.random() < 0.5:
        rl = sc.ref_line_straight(x0=float(rng.uniform(-300, 300)), y0=float(rng.uniform(-300, 300)),
                                  heading=float(rng.uniform(-np.pi, np.pi)), semantic_range=(100, 140))
    else:
        rl = sc.ref_line_arc(radius=float(rng.uniform(60, 400)) * float(rng.choice([1.0, 1.0, 1.0])),
                             heading0=float(rng.uniform(-np.pi, np.pi)), semantic_range=(90, 130))
    step = 0.5
    ts = np.array([0.0] + [dt + i * dt for i in range(T - 1)])
    objs = []
    for k in range(int(rng.integers(0, 15))):
        still = bool(rng.random() < 0.3)
        lc = float(rng.uniform(-4.0, 4.0))
        objs.append((sc.prediction_soup(rl, step, float(rng.uniform(5.0, 190.0)), 0.0 if still else float(rng.uniform(0.2, 12.0)),
                                        float(rng.uniform(2.0, 12.0)), lc - float(rng.uniform(0.3, 1.5)),
                                        lc + float(rng.uniform(0.3, 1.5)), ts), still))
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(rl, step)
    o = oe.OracleEnv(ep)
    o.set_ref_line(rl, step)
    for g, st in objs:
        env.insert_geometry(g, st)
        o.insert_geometry(g, st)
    env.update()
    img, occ, dl = o.update()
    assert np.array_equal(env.get_occ_map_cartesian(), img), f"seed {seed}: occupancy image"
    assert np.array_equal(env.get_occ_map(), occ), f"seed {seed}: Frenet occupancy"
    assert np.array_equal(env.get_dist_map_lon(), dl), f"seed {seed}: distance map"

    # lateral lattice
    lp = tpl_amd.PolyLatPlannerParams()
    lp.s_max = min(float(ep.s_max), float(rng.choice([100.0, 150.0])))
    lp.s_steps = int(lp.s_max) + 1
    lp.width_veh, lp.length_veh = 1.9, 4.9
    lp.l_trg = float(rng.uniform(-1.0, 1.0))
    start = tpl_amd.PolyLatTrajPoint()
    start.s, start.l, start.dl, start.v = float(rng.uniform(0, 20)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-0.1, 0.1)), float(rng.uniform(2, 14))
    pl = tpl_amd.PolyLatPlanner()
    pl.reinit_buffers(lp)
    traj = pl.update(start, env)
    nodes, best, pts, q = opl.plan(lp, o.c, start, o.ref, o.step, o.offsets, occ)
    got = pl.nodes()
    assert np.allclose(got[..., 1:, :2], nodes[..., 1:, :2], rtol=1e-5, atol=1e-6), f"seed {seed}: lattice values"
    assert np.array_equal(got[..., 2], nodes[..., 2]), f"seed {seed}: collision stations"
    assert pl.best == best, f"seed {seed}: selected lattice path"
    # the final trajectory is host arithmetic on both sides, but of two compilers (sin / cos vs sincos): last-bit differences
    assert np.allclose(traj.as_numpy(), pts, rtol=0, atol=1e-12), f"seed {seed}: lattice trajectory"

    # (s, ds, l) planner
    qp = tpl_amd.DynProgLatLonPlannerParams()
    qp.s_steps, qp.ds_steps, qp.l_steps, qp.t_steps = int(rng.choice([64, 101])), int(rng.integers(8, 20)), int(rng.choice([7, 11, 15])), int(rng.integers(3, 8))
    qp.ds_max = float(rng.choice([15.0, 18.0]))
    qp.dt = dt
    qp.dt_start = float(rng.choice([dt, 0.4 * dt]))
    qp.length_veh, qp.width_veh = 4.9, 1.9
    qp.l_trg = float(rng.uniform(-0.5, 0.5))
    st = tpl_amd.LatLonState()
    st.s, st.ds, st.l = float(rng.uniform(0, 30)), float(rng.uniform(0, 14)), float(rng.uniform(-1.5, 1.5))
    ll = tpl_amd.DynProgLatLonPlanner()
    ll.reinit_buffers(qp)
    rule = None
    if rules:
        rule = (int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 4)))
        ll.set_filter_rule(*rule)
        oll.set_rule(*rule)
    ll.traj_dp[0] = st
    ll.update_traj_dp(env)
    try:
        tables, otraj = oll.update_traj_dp(qp, o.c, dl, o.ref, o.step, st)
    finally:
        if rules:
            oll.set_rule()
    for t in range(1, qp.t_steps):
        g = ll.copy_backward_pass_tex(t)
        assert np.array_equal(g.view(np.uint32), tables[t].view(np.uint32)), f"seed {seed}: (s,ds,l) stage {t}"
    assert bytes(ll.traj_dp._as_array()) == bytes(otraj), f"seed {seed}: (s,ds,l) trajectory"
    return int(np.count_nonzero(occ == 1.0)), len(objs)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rules = "--rules" in sys.argv
    n = int(args[0]) if len(args) > 0 else 100
    first = int(args[1]) if len(args) > 1 else 0
    cells = objs = 0
    for seed in range(first, first + n):
        c, k = run_scene(seed, rules)
        cells += c
        objs += k
    print(f"{n} random scenes ({objs} objects, {cells} occupied Frenet cells): occupancy pipeline and (s,ds,l) planner "
          f"{'(under random texture-filter rules) ' if rules else ''}bit-identical, lateral lattice within 1e-5 with identical discrete outcomes")


if __name__ == "__main__":
    main()
