"""DP (s, ds, l) lateral/longitudinal planner: tpl_amd.DynProgLatLonPlanner on the
GPU against oracle/lat_lon_oracle.c through the C-ABI.  The backward tables are point sampled and
the arithmetic is +, -, *, /, min, max on floats: tables, the DP trajectory and the re-evaluated
trajectory must be bit-identical.  The smoothing pass is fp64 linear algebra written twice
(product, oracle) and compared at 1e-9."""
import ctypes as C

import numpy as np
import pytest

import oracle_env as oe
import oracle_lat_lon as oll
import tpl_amd
from benchkit import scenarios as sc
from tpl_amd._capi import TplLatLonCartState, TplLatLonState

pytestmark = pytest.mark.gpu


def _env(kind, seed, ep=None, n_objects=6, semantic=True):
    ep = ep or sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = sc.env_scene(kind, seed=seed, n_objects=n_objects)
    if not semantic:
        rl[:, 8] = 0.0
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(rl, step)
    o = oe.OracleEnv(ep)
    o.set_ref_line(rl, step)
    for g, st in objs:
        env.insert_geometry(g, st)
        o.insert_geometry(g, st)
    env.update()
    _, occ, dist_lon = o.update()
    assert np.array_equal(env.get_dist_map_lon(), dist_lon)
    return env, o, dist_lon


def _params(**kw):
    p = tpl_amd.DynProgLatLonPlannerParams()
    p.s_steps, p.ds_steps, p.l_steps, p.t_steps = 101, 19, 11, 7
    p.ds_max = 18.0
    p.length_veh, p.width_veh = 4.9, 1.9
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _start(s=2.0, ds=7.0, l=0.3):
    st = tpl_amd.LatLonState()
    st.s, st.ds, st.l = s, ds, l
    return st


def _plan_both(env, o, dist_lon, p, start):
    pl = tpl_amd.DynProgLatLonPlanner()
    pl.reinit_buffers(p)
    pl.traj_dp[0] = start
    pl.update_traj_dp(env)
    tables, otraj = oll.update_traj_dp(p, o.c, dist_lon, o.ref, o.step, start)
    return pl, tables, otraj


def _assert_tables(pl, tables, T):
    for t in range(1, T):
        got = pl.copy_backward_pass_tex(t)
        assert np.array_equal(got.view(np.uint32), tables[t].view(np.uint32)), \
            f"stage {t}: {np.count_nonzero(got != tables[t])} of {got.size} values differ"


@pytest.mark.parametrize("kind,seed", [("straight", 0), ("arc", 1)])
def test_tables_and_dp_trajectory_bit_exact(kind, seed):
    env, o, dist_lon = _env(kind, seed)
    p = _params()
    pl, tables, otraj = _plan_both(env, o, dist_lon, p, _start())
    _assert_tables(pl, tables, p.t_steps)
    got = pl.traj_dp._as_array()
    assert bytes(got) == bytes(otraj)
    a = pl.traj_dp.as_numpy()
    assert a[-1, 1] > a[0, 1] and np.all(np.diff(a[:, 0]) > 0)            # moves forward in time and station


def test_other_grid_weights_and_start_states():
    env, o, dist_lon = _env("straight", 4, n_objects=9)
    p = _params(s_steps=64, ds_steps=12, l_steps=9, t_steps=5, dt=0.8, dt_start=0.4, w_lat_dist=2.0, d_lat_comf=1.0,
                l_trg=-0.5, t_st_min=2.0, t_st_max=1.0, s_st=40.0, w_xing_slow=5.0, slope_abs_max=0.3)
    for st in (_start(0.0, 0.0, -1.0), _start(30.0, 15.0, 1.5)):
        pl, tables, otraj = _plan_both(env, o, dist_lon, p, st)
        _assert_tables(pl, tables, p.t_steps)
        assert bytes(pl.traj_dp._as_array()) == bytes(otraj)


def test_query_is_the_point_sampled_cell():
    env, o, dist_lon = _env("straight", 0)
    p = _params()
    pl, tables, _ = _plan_both(env, o, dist_lon, p, _start())
    ss = (p.s_max - p.s_min) / (p.s_steps - 1)
    sd = (p.ds_max - p.ds_min) / (p.ds_steps - 1)
    sl = (p.l_max - p.l_min) / (p.l_steps - 1)
    for (is_, id_, il) in [(0, 0, 0), (50, 9, 5), (100, 18, 10)]:
        q = pl.query_backward_pass_tex(3, p.s_min + is_ * ss + 0.3 * ss, p.ds_min + id_ * sd - 0.3 * sd,
                                       p.l_min + il * sl + 0.2 * sl)
        assert np.array_equal(q, tables[3, il, id_, is_])


def test_reeval_smooth_and_cartesian():
    env, o, dist_lon = _env("arc", 1)
    p = _params()
    pl, tables, otraj = _plan_both(env, o, dist_lon, p, _start())
    # reevalTraj of the DP trajectory
    re = pl.reeval_traj(pl.traj_dp, env)
    ore = oll.reeval(p, o.c, dist_lon, o.ref, o.step, otraj)
    assert bytes(re._as_array()) == bytes(ore)
    # smoothing
    pl.traj_smooth[0] = _start()
    pl.update_traj_smooth(env)
    out = (TplLatLonState * 256)()
    n = oll.lib().llo_update_traj_smooth(C.byref(p._c), otraj, len(otraj), C.byref(_start()._c), out, 256)
    assert n == len(pl.traj_smooth.states) and n > 10
    assert np.allclose(pl.traj_smooth.as_numpy_full(), oll.states_np(out[:n]), rtol=1e-9, atol=1e-9)
    # the reference's array conversions (bindings_dp_lat_lon.cu:56-70, 132-184, 32-44, 83-116): column subsets
    assert pl.traj_smooth.as_numpy().shape == (n, 9) and pl.traj_smooth[0].as_numpy().shape == (7,)
    assert np.array_equal(pl.traj_smooth.as_numpy(), pl.traj_smooth.as_numpy_full()[:, :9])
    assert pl.traj_smooth.state(np.array([0.1, 0.2])).shape == (2, 7) and pl.traj_smooth.lerp(np.array([0.1])).shape == (1, 7)
    assert len(pl.lat_dists) == 1 and pl.lat_dists[0].d_left == 0.0          # lat_lon_planner.cu:729-747
    # Frenet -> Cartesian (host arithmetic on both sides: identical), called the way
    # dp_lat_lon_planner.py:148 calls it: with the environment's RefLine
    pl.update_traj_cart(env.ref_line)
    ocart = (TplLatLonCartState * len(otraj))()
    oll.lib().llo_frenet_to_cartesian(o.ref.ctypes.data_as(C.POINTER(C.c_double)), len(o.ref), o.step,
                                      o.offsets.ctypes.data_as(C.POINTER(C.c_double)), otraj, len(otraj), ocart)
    assert bytes(pl.traj_dp_cart._as_array()) == bytes(ocart)
    assert len(pl.traj_smooth_cart.states) == n
    c = pl.traj_smooth_cart.lerp(0.55)
    assert pl.traj_smooth_cart.as_numpy().shape == (n, 8) and c.as_numpy().shape == (8,)
    assert pl.traj_smooth_cart.lerp(np.array([0.5, 1.0])).shape == (2, 7)
    assert np.isfinite(c.x) and np.isfinite(c.heading)
    # the environment itself and a free-standing RefLine with the same points give the same conversion
    via_env = pl.frenet_to_cartesian(pl.traj_dp, env)
    rl = env.ref_line
    free = tpl_amd.RefLine(rl.as_array().copy(), rl.step_size, rl.x_offset, rl.y_offset)
    via_free = pl.frenet_to_cartesian(pl.traj_dp, free)
    assert bytes(via_env._as_array()) == bytes(ocart)
    assert np.allclose(via_free.as_numpy(), pl.traj_dp_cart.as_numpy(), rtol=0, atol=1e-9)
    with pytest.raises(TypeError):
        pl.frenet_to_cartesian(pl.traj_dp, 3)


def test_free_road_plan_cruises_in_lane_and_flags():
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl = sc.ref_line_straight(x0=0.0, y0=0.0, heading=0.0, v_max=10.0)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(rl, 0.5)
    env.update()
    p = _params()
    pl = tpl_amd.DynProgLatLonPlanner()
    pl.reinit_buffers(p)
    pl.traj_dp[0] = _start(0.0, 8.0, 0.0)
    pl.update_traj_dp(env)
    a = pl.traj_dp.as_numpy_full()
    assert np.all(a[:, 10] == 0.0) and np.all(a[:, 11] == 0)               # no constraint violated
    assert np.all(np.abs(a[:, 5]) < 1.0) and np.all(a[:, 2] <= 10.0 + 1e-6)   # stays near l_trg, below v_max


def test_errors():
    pl = tpl_amd.DynProgLatLonPlanner()
    env = tpl_amd.DynProgEnvironment()
    with pytest.raises(RuntimeError, match="never updated"):
        pl.update_traj_dp(env)
    bad = tpl_amd.DynProgLatLonPlannerParams()
    bad.l_steps = 1
    with pytest.raises(ValueError):
        pl.reinit_buffers(bad)


@pytest.mark.parametrize("quant,lerp,order", [(2, 0, 0), (1, 1, 0), (1, 0, 1), (1, 0, 2), (1, 0, 3), (2, 2, 1), (0, 1, 2)])
def test_filter_rule_variants_match_the_oracle_under_the_same_rule(quant, lerp, order):
    """The texture filter's frozen arithmetic -- weight rule, blend form, blend order / eight-corner sums -- as a run-time choice of
    this planner too (tplhip_lat_lon_set_filter_rule; HP-1: tests/test_dp_gpu.py).  It reaches the forward pass' filtered fetch of
    the value tables, the distance-map and the reference-line look-ups; each variant must reproduce
    oracle/lat_lon_oracle.c under the same rule (llo_set_rule: one library, the rule is run-time state) bit for bit:
    tables, the DP trajectory, the re-evaluated trajectory."""
    env, o, dist_lon = _env("arc", 1)
    p = _params()
    start = _start(l=0.37)
    with oll.rule(quant, lerp, order):
        tables, otraj = oll.update_traj_dp(p, o.c, dist_lon, o.ref, o.step, start)
        ore = oll.reeval(p, o.c, dist_lon, o.ref, o.step, otraj)
    tables0, otraj0 = oll.update_traj_dp(p, o.c, dist_lon, o.ref, o.step, start)
    pl = tpl_amd.DynProgLatLonPlanner()
    pl.reinit_buffers(p)
    pl.set_filter_rule(quant, lerp, order)
    pl.traj_dp[0] = start
    pl.update_traj_dp(env)
    _assert_tables(pl, tables, p.t_steps)
    assert bytes(pl.traj_dp._as_array()) == bytes(otraj)
    re = pl.reeval_traj(pl.traj_dp, env)
    assert bytes(re._as_array()) == bytes(ore)
    # (the backward tables are point sampled: only the prelude's filtered distances and reference-line values reach them, and on
    # this scene several variants give the default's bits; truncated weights do not)
    if (quant, lerp, order) == (2, 0, 0):
        assert bytes(otraj) != bytes(otraj0) or any(not np.array_equal(tables[t], tables0[t]) for t in range(1, p.t_steps))
    # the default comes back
    pl.set_filter_rule(1, 0, 0)
    pl.traj_dp[0] = start
    pl.update_traj_dp(env)
    assert bytes(pl.traj_dp._as_array()) == bytes(otraj0)
    with pytest.raises(ValueError):
        pl.set_filter_rule(1, 0, 4)
