"""Lateral quintic-lattice planner: tpl_amd.PolyLatPlanner on the GPU against
oracle/poly_lat_oracle.c through the C-ABI.

Tolerances: the lattice runs float atan / cos / sqrt on the device; the device library and the
C library agree to about one unit in the last place, not bit for bit, so node values are
compared at 1e-5 relative (+1e-6 absolute).  The local cost adds 10e6 * constraint violation
(poly_lat_planner.cu:89): one float ulp of an O(1..10) constraint term is worth up to ~10 cost
units, so costs get an absolute allowance of 32 (node) and 32 per station (path sum) on top of
1e-5 relative.  Everything discrete -- collision stations, the selected path -- and the final
trajectory (host arithmetic on both sides) must be identical."""
import ctypes as C

import numpy as np
import pytest

import oracle_env as oe
import oracle_poly_lat as opl
import tpl_amd
from benchkit import scenarios as sc

pytestmark = pytest.mark.gpu


def _setup(kind, seed, ep=None, n_objects=6):
    ep = ep or sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = sc.env_scene(kind, seed=seed, n_objects=n_objects)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(rl, step)
    o = oe.OracleEnv(ep)
    o.set_ref_line(rl, step)
    for g, st in objs:
        env.insert_geometry(g, st)
        o.insert_geometry(g, st)
    env.update()
    _, occ, _ = o.update()
    assert np.array_equal(env.get_occ_map(), occ)
    return env, o, occ


def _start(s=5.0, l=0.4, dl=0.05, ddl=0.0, v=8.0):
    st = tpl_amd.PolyLatTrajPoint()
    st.s, st.l, st.dl, st.ddl, st.v = s, l, dl, ddl, v
    return st


def _check(env, o, occ, p, start):
    pl = tpl_amd.PolyLatPlanner()
    pl.reinit_buffers(p)
    traj = pl.update(start, env)
    nodes, best, pts, q = opl.plan(p, o.c, start, o.ref, o.step, o.offsets, occ)
    got = pl.nodes()
    # l, v, local cost
    for ch in (0, 1):
        assert np.allclose(got[..., 1:, ch], nodes[..., 1:, ch], rtol=1e-5, atol=1e-6), f"channel {ch}"
    assert np.allclose(got[..., 1:, 3], nodes[..., 1:, 3], rtol=1e-5, atol=32.0)
    assert np.allclose(got[..., 0, :2], nodes[..., 0, :2], rtol=1e-5, atol=1e-6)
    # collision stations are discrete: identical (also the per-path minimum kept in node 0)
    assert np.array_equal(got[..., 2], nodes[..., 2])
    assert np.allclose(got[..., 0, 3], nodes[..., 0, 3], rtol=1e-5, atol=32.0 * p.s_steps)       # path costs
    assert pl.best == best
    assert np.array_equal(traj.as_numpy(), pts)
    assert np.array_equal(np.array([traj.poly._c.x0, traj.poly._c.x1, traj.poly._c.d, *traj.poly._c.c]),
                          np.array([q.x0, q.x1, q.d, *q.c]))
    return pl, traj, nodes, best


@pytest.mark.parametrize("kind,seed", [("straight", 0), ("arc", 1), ("arc", 5)])
def test_lattice_and_selection_parity(kind, seed):
    env, o, occ = _setup(kind, seed)
    p = tpl_amd.PolyLatPlannerParams()
    p.s_max, p.s_steps = 150.0, 151
    p.width_veh, p.length_veh = 1.9, 4.9
    pl, traj, nodes, best = _check(env, o, occ, p, _start())
    assert len(traj.points) == 151
    assert (nodes[..., 0, 2] < 1000.0).any()          # these scenes do block paths


def test_other_lattice_and_start_states():
    env, o, occ = _setup("straight", 9, n_objects=10)
    p = tpl_amd.PolyLatPlannerParams()
    p.s_max, p.s_steps = 120.0, 97
    p.l_dst_min, p.l_dst_max, p.l_dst_steps = -3.0, 3.0, 13
    p.s_dst_min, p.s_dst_max, p.s_dst_steps = 15.0, 60.0, 7
    p.w_right, p.w_k, p.w_dl, p.w_ddl, p.l_trg = 2.0, 0.3, 0.5, 0.2, -0.5
    for st in (_start(0.0, -1.2, -0.1, 0.02, 3.0), _start(30.0, 2.0, 0.0, 0.0, 14.0)):
        _check(env, o, occ, p, st)


def test_free_road_keeps_the_target_offset_and_planner_pair_runs():
    """No obstacles: the cheapest path ends on l_trg; its points feed the DP (s,v,a) planner the way
    poly_lat_dp_lon_planner.py:139-175 chains the two."""
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl = sc.ref_line_straight(x0=0.0, y0=0.0, heading=0.0)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(rl, 0.5)
    env.update()
    o = oe.OracleEnv(ep)
    o.set_ref_line(rl, 0.5)
    _, occ, _ = o.update()
    p = tpl_amd.PolyLatPlannerParams()
    pl, traj, nodes, best = _check(env, o, occ, p, _start(0.0, 0.8, 0.0, 0.0, 8.0))
    l_dst = p.l_dst_min + best[0] * (p.l_dst_max - p.l_dst_min) / (p.l_dst_steps - 1)
    assert l_dst == pytest.approx(0.0, abs=1e-6)
    # path for the longitudinal planner: PathState{x, y, s, l, k, v_max, distance} at 1 m steps
    lp = sc.lon_params(s_steps=64, v_steps=24, a_steps=12, t_steps=10)
    samples = traj.lerp(np.arange(lp.path_steps) * lp.path_step_size)
    path = np.stack([samples[:, 7], samples[:, 8], samples[:, 5], samples[:, 1], samples[:, 11], samples[:, 6],
                     samples[:, 10]], axis=1).astype(np.float32)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(lp)
    lon = dp.update(sc.initial_state(v=8.0), path, env)
    assert len(lon.states) == 10 and lon.states[-1].s > 30.0


def test_errors():
    pl = tpl_amd.PolyLatPlanner()
    env = tpl_amd.DynProgEnvironment()
    with pytest.raises(RuntimeError, match="reinit_buffers was never called on the lateral planner"):
        pl.update(_start(), env)
    pl.reinit_buffers(tpl_amd.PolyLatPlannerParams())
    with pytest.raises(RuntimeError, match="no reference line"):
        pl.update(_start(), env)
    bad = tpl_amd.PolyLatPlannerParams()
    bad.s_steps = 5000
    with pytest.raises(ValueError):
        pl.reinit_buffers(bad)
