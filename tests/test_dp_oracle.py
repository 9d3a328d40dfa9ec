"""CPU tests of the HP-1 oracle (oracle/dp_lon_oracle.c).

The reference holds no golden vector for this path (parity
unpinned), so the oracle is pinned with hand-derivable known-answer tests:
closed-form dynamics, interpolation identities, the texture-filter rule, wall /
empty-road scenarios and the tie-breaking rule of the lexicographic argmin.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_dp as O
from benchkit import scenarios as sc
from tpl_amd._capi import TplLonState, TplPathState


def _state(t=0.0, s=0.0, v=0.0, a=0.0, j=0.0):
    st = TplLonState()
    st.t, st.s, st.v, st.a, st.j = t, s, v, a, j
    return st


def test_lon_dynamics_closed_form():
    # lon_planner.cu:240-250: constant-jerk step, s and v clipped from below
    x0 = _state(t=1.0, s=10.0, v=8.0, a=1.0)
    out = TplLonState()
    O.lib().dpo_lon_dynamics(C.byref(x0), 0.5, 2.0, C.byref(out))
    assert out.t == pytest.approx(3.0)
    assert out.s == pytest.approx(10.0 + 8.0 * 2 + 0.5 * 1.0 * 4 + 0.5 / 6 * 8, rel=1e-6)
    assert out.v == pytest.approx(8.0 + 1.0 * 2 + 0.5 * 0.5 * 4, rel=1e-6)
    assert out.a == pytest.approx(2.0)
    assert out.j == 0.5
    assert out.cost == 0.0 and out.constr == 0.0
    # braking past standstill: v clips at 0, s never decreases
    x0 = _state(s=5.0, v=1.0, a=-3.0)
    O.lib().dpo_lon_dynamics(C.byref(x0), -3.0, 2.0, C.byref(out))
    assert out.v == 0.0
    assert out.s == 5.0


def test_lon_dynamics_mixed_precision():
    # the double literals of the reference promote the a/j terms: check one case
    # against an explicit numpy restatement of the promotion rules (one rounding per
    # operation: the host instantiation, which no rule R1 reaches)
    x0 = _state(s=np.float32(3.3), v=np.float32(7.7), a=np.float32(-1.1))
    j, dt = np.float32(0.7), np.float32(0.9)
    out = TplLonState()
    O.lib().dpo_lon_dynamics_host(C.byref(x0), j, dt, C.byref(out))
    s, v, a = np.float32(3.3), np.float32(7.7), np.float32(-1.1)
    sf = np.float32(s + np.float32(v * dt))
    sd = np.float64(sf) + ((0.5 * np.float64(a)) * np.float64(dt)) * np.float64(dt)
    sd = sd + ((((1.0 / 6.0) * np.float64(j)) * np.float64(dt)) * np.float64(dt)) * np.float64(dt)
    assert out.s == max(s, np.float32(sd))
    vf = np.float32(v + np.float32(a * dt))
    vd = np.float64(vf) + ((0.5 * np.float64(j)) * np.float64(dt)) * np.float64(dt)
    assert out.v == max(np.float32(0), np.float32(vd))


def test_interp_path_identities():
    path = sc.straight_path(200, 1.0)
    res = TplPathState()
    # exactly on a sample
    O.lib().dpo_interp_path(O.fptr(path), 57.0, 1.0, 200, C.byref(res))
    assert res.s == 57.0 and res.v_max == path[57, 5]
    # half way
    O.lib().dpo_interp_path(O.fptr(path), 57.5, 1.0, 200, C.byref(res))
    assert res.s == 57.5
    assert res.v_max == np.float32(np.float32(path[57, 5] * np.float32(0.5)) + np.float32(path[58, 5] * np.float32(0.5)))
    # clamps below and above (common.cuh:124-125)
    O.lib().dpo_interp_path(O.fptr(path), -3.0, 1.0, 200, C.byref(res))
    # i0 = i1 = 0, alpha = -3 -> extrapolates with weights (4, -3) of the same sample
    assert res.v_max == pytest.approx(path[0, 5], rel=1e-6)
    O.lib().dpo_interp_path(O.fptr(path), 1e9, 1.0, 200, C.byref(res))
    assert np.isfinite(res.v_max)


def _small():
    p = sc.lon_params(s_steps=33, v_steps=9, a_steps=7, t_steps=5)
    ep = sc.env_params(t_steps=5, s_steps=41, l_steps=11)
    return O.cparams(p), O.cparams(ep)


def test_read_node_grid_points_and_quantisation():
    p, _ = _small()
    rng = np.random.default_rng(0)
    table = rng.uniform(0, 10, size=(p.a_steps, p.v_steps, p.s_steps, 4)).astype(np.float32)
    out = np.zeros(4, dtype=np.float32)
    # exactly on grid points the filter returns the texel (weights 0)
    for (ia, iv, i_s) in [(0, 0, 0), (3, 4, 17), (6, 8, 32)]:
        s = p.s_min + i_s * p._s_step_size
        v = p.v_min + iv * p._v_step_size
        a = p.a_min + ia * p._a_step_size
        O.lib().dpo_read_node(C.byref(p), O.fptr(table), s, v, a, O.fptr(out))
        assert np.allclose(out[:3], table[ia, iv, i_s, :3], rtol=2e-6)
        assert out[3] == 0.0
    # rule R2: weights have 8 fractional bits -> a station 1/1024 of a cell past a
    # grid point still returns the grid value, 1/256 moves it by exactly one step
    s0 = p.s_min + 10 * p._s_step_size
    v0, a0 = p.v_min, p.a_min
    O.lib().dpo_read_node(C.byref(p), O.fptr(table), s0 + p._s_step_size / 1024, v0, a0, O.fptr(out))
    assert out[0] == pytest.approx(table[0, 0, 10, 0], rel=1e-6)
    O.lib().dpo_read_node(C.byref(p), O.fptr(table), s0 + p._s_step_size * 0.5, v0, a0, O.fptr(out))
    assert out[0] == pytest.approx(0.5 * (table[0, 0, 10, 0] + table[0, 0, 11, 0]), rel=1e-5)
    # clamp addressing outside the grid
    O.lib().dpo_read_node(C.byref(p), O.fptr(table), -50.0, -5.0, -9.0, O.fptr(out))
    assert np.allclose(out[:3], table[0, 0, 0, :3], rtol=2e-6)
    O.lib().dpo_read_node(C.byref(p), O.fptr(table), 1e4, 1e3, 99.0, O.fptr(out))
    assert np.allclose(out[:3], table[-1, -1, -1, :3], rtol=2e-6)


def test_blend_with_weight_zero_or_twin_rows_is_what_the_single_a_sweeps_compute():
    """What ACT_SINGLE_A of the plain-fp32 sweeps rests on (tpl_amd/csrc/dp_lon_internal.h): the reference's blend
    lerp(p, q, w) = fma(q, w, p * (1 - w)) returns p bit for bit when w == 0, whatever finite q is -- table values are
    finite and never -0 -- and a blend of a row with itself needs only that row.  Checked in float32 on the oracle's
    filter through a table whose two a-rows differ wildly: on an a-grid point the upper row does not matter."""
    p, _ = _small()
    rng = np.random.default_rng(7)
    table = rng.uniform(0, 10, size=(p.a_steps, p.v_steps, p.s_steps, 4)).astype(np.float32)
    table[..., 2] = rng.uniform(-3, 2, size=table.shape[:3]).astype(np.float32)       # the jerk channel may be negative
    table[0, 0, 0, :3] = 0.0                                                            # ... and a cell may be all +0
    twin = table.copy()
    twin[1:] = rng.uniform(-1e6, 1e6, size=twin[1:].shape).astype(np.float32)           # every row above a-row 0: garbage
    out, out2 = np.zeros(4, dtype=np.float32), np.zeros(4, dtype=np.float32)
    for iv in (0, 3, p.v_steps - 1):
        for i_s in (0, 5, p.s_steps - 1):
            s = p.s_min + i_s * p._s_step_size
            v = p.v_min + iv * p._v_step_size
            O.lib().dpo_read_node(C.byref(p), O.fptr(table), s, v, p.a_min, O.fptr(out))        # a on the first a-row: w_a == 0
            O.lib().dpo_read_node(C.byref(p), O.fptr(twin), s, v, p.a_min, O.fptr(out2))
            assert np.array_equal(out.view(np.uint32), out2.view(np.uint32)), (iv, i_s, out, out2)
            O.lib().dpo_read_node(C.byref(p), O.fptr(table), s, v, p.a_min - 7.0, O.fptr(out))  # clamped below: both taps are a-row 0
            O.lib().dpo_read_node(C.byref(p), O.fptr(twin), s, v, p.a_min - 7.0, O.fptr(out2))
            assert np.array_equal(out.view(np.uint32), out2.view(np.uint32)), (iv, i_s, out, out2)
    # the identity itself, on random bit patterns of the admissible kind
    pp = np.abs(rng.standard_normal(100000).astype(np.float32)) * np.float32(50.0)
    pp[:100] = 0.0
    qq = (rng.standard_normal(100000) * 1e5).astype(np.float32)
    w = np.float32(0.0)
    got = (qq.astype(np.float64) * float(w) + (pp * (np.float32(1.0) - w)).astype(np.float64)).astype(np.float32)   # fma: one rounding
    assert np.array_equal(got.view(np.uint32), pp.view(np.uint32))


def test_dist_map_path_scan():
    p, ep = _small()
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = np.zeros((ep.l_steps, ep.s_steps, ep.t_steps), dtype=np.float32)
    il = int((0.0 - ep.l_min) / ep._l_step_size)
    occ[il, 20, :] = 1.0
    occ[il, 30, 2] = 3.0
    dist = O.dist_map_path(ep, occ, path, p.path_steps, p.path_step_size)
    ds = np.float32(ep._s_step_size)
    # behind the last obstacle the scan counts up from 10000 (env.cu:142)
    assert dist[0, ep.s_steps - 1] == np.float32(np.float32(10000.0) + ds)
    assert dist[0, 20] == 0.0
    assert dist[0, 19] == ds
    assert dist[0, 18] == np.float32(ds + ds)
    assert dist[2, 30] == 0.0 and dist[2, 29] == ds and dist[2, 20] == 0.0
    assert dist[1, 30] != 0.0
    # sequential float accumulation, not idx * ds
    d = np.float32(0.0)
    for _ in range(20):
        d = np.float32(d + ds)
    assert dist[0, 0] == d
    # lookup: exact at cell centres, linear in between, slice picked by env.cu:258
    f = O.lib().dpo_interp_dist_map_path
    s19 = ep.s_min + 19 * ep._s_step_size
    assert f(C.byref(ep), O.fptr(dist), 0.0, s19) == pytest.approx(ds, rel=1e-6)
    assert f(C.byref(ep), O.fptr(dist), 2.0, ep.s_min + 29.5 * ep._s_step_size) == pytest.approx(0.5 * ds, rel=1e-3)
    # t = dt_start + dt -> slice 2
    assert f(C.byref(ep), O.fptr(dist), 2.0, ep.s_min + 30 * ep._s_step_size) == pytest.approx(0.0, abs=1e-6)


def test_empty_road_cruise_and_terminal_stage():
    # KAT (i): no obstacles, start at the speed limit with a = 0: the cheapest
    # plan never violates a constraint and keeps |j| small
    p, ep = _small()
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = np.zeros((ep.l_steps, ep.s_steps, ep.t_steps), dtype=np.float32)
    tables, trajs, _ = O.update(p, ep, path, occ, [sc.initial_state(0.0, 8.0, 0.0)])
    T = p.t_steps
    # terminal stage stores (state_cost, 0, 0, 0) everywhere (lon_planner.cu:93-97)
    assert np.all(tables[T - 1, ..., 1:] == 0.0)
    assert np.all(tables[T - 1, ..., 0] > 0.0)
    # stage 0 is never written (lon_planner.cu:347)
    assert np.all(tables[0] == 0.0)
    tr = trajs[0]
    assert np.all(tr[1:-1, 6] < 0.1)                 # valid plan predicate of the caller
    assert np.all(np.diff(tr[:, 1]) >= 0.0)          # stations never decrease
    assert np.allclose(np.diff(tr[:, 0]), 1.0)       # t advances by dt


def test_wall_raises_constraint():
    # KAT (ii): a wall right ahead: every action from a fast cell overshoots s_dist
    p, ep = _small()
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = np.zeros((ep.l_steps, ep.s_steps, ep.t_steps), dtype=np.float32)
    occ[:, 8:, :] = 1.0                               # wall from s = 40 m on
    dist = O.dist_map_path(ep, occ, path, p.path_steps, p.path_step_size)
    T = p.t_steps
    tf = int(O.lib().dpo_table_floats(C.byref(p)))
    term = np.zeros(tf, dtype=np.float32)
    O.lib().dpo_backward_stage(C.byref(p), C.byref(ep), T - 1, O.fptr(term), O.fptr(term),
                               O.fptr(path), O.fptr(dist), 1)
    stage = np.zeros(tf, dtype=np.float32)
    O.lib().dpo_backward_stage(C.byref(p), C.byref(ep), T - 2, O.fptr(term), O.fptr(stage),
                               O.fptr(path), O.fptr(dist), 1)
    stage = stage.reshape(p.a_steps, p.v_steps, p.s_steps, 4)
    ia0 = int(round((0.0 - p.a_min) / p._a_step_size))
    # a cell 5 m in front of the wall driving 18 m/s cannot stop: constr > 0
    i_s = int(round(35.0 / p._s_step_size))
    iv = int(round(18.0 / p._v_step_size))
    assert stage[ia0, iv, i_s, 1] > 1.0
    # a standing cell far from the wall has a constraint-free action
    assert stage[ia0, 0, 0, 1] == 0.0


def test_tie_break_lowest_action():
    # KAT (iv): with all weights zero every action costs the same -> index 0 wins
    p, ep = _small()
    p.w_progress = p.w_a = p.w_j = p.w_snap = p.w_safety_dist = 0.0
    path = sc.straight_path(p.path_steps, p.path_step_size)
    path[:, 5] = 1000.0                                # no speed limit
    dist = np.full((ep.t_steps, ep.s_steps), 10000.0, dtype=np.float32)
    table = np.zeros((p.a_steps, p.v_steps, p.s_steps, 4), dtype=np.float32)
    tp = _state(t=1.0, s=10.0, v=5.0, a=0.0)
    p.a_min, p.a_max = -100.0, 100.0                   # no acceleration constraint either
    O.lib().dpo_update_step_sizes(C.byref(p))
    O.lib().dpo_eval_node(C.byref(p), C.byref(ep), 9, C.byref(tp), O.fptr(table), O.fptr(path),
                          O.fptr(dist), 1.0, 1, 0)
    assert tp.j == p.j_min and tp.cost == 0.0 and tp.constr == 0.0
    # constr has priority over cost: make action 0 violate, the next one wins
    p.a_min = -2.5                                     # a' = 0 + j*dt < a_min only for j = -3
    O.lib().dpo_update_step_sizes(C.byref(p))
    table = np.zeros((p.a_steps, p.v_steps, p.s_steps, 4), dtype=np.float32)
    tp = _state(t=1.0, s=10.0, v=5.0, a=0.0)
    O.lib().dpo_eval_node(C.byref(p), C.byref(ep), 9, C.byref(tp), O.fptr(table), O.fptr(path),
                          O.fptr(dist), 1.0, 1, 0)
    j1 = np.float32(p.j_min) + np.float32(p.j_max - p.j_min) * (np.float32(1) / np.float32(8))
    assert tp.j == j1 and tp.constr == 0.0


def test_traj_state_brackets():
    # LonTraj::state, lon_planner.cu:252-260 with InterpVars utils.hpp:59-78
    n = 4
    arr = (TplLonState * n)()
    for i in range(n):
        arr[i].t = float(i)
        arr[i].s = 10.0 * i
        arr[i].v = 10.0
        arr[i].j = 0.0
    out = TplLonState()
    assert O.lib().dpo_traj_state(arr, n, 1.5, C.byref(out)) == 0
    assert out.s == pytest.approx(15.0)
    O.lib().dpo_traj_state(arr, n, -1.0, C.byref(out))      # before the first: extrapolate from state 0
    assert out.s == pytest.approx(0.0)                      # s is clipped from below by lonDynamics
    O.lib().dpo_traj_state(arr, n, 5.0, C.byref(out))       # past the end: from state n-2
    assert out.s == pytest.approx(20.0 + 30.0)


def test_product_host_helpers_match_oracle():
    # tplhip_lon_dynamics / tplhip_lon_traj_state run on the host inside libtplhip.so: the HOST instantiation of the
    # reference's __interop__ lonDynamics (compiled for baseline x86-64: no fused multiply-add, whatever R1 says)
    import tpl_amd
    from tpl_amd import _capi
    lib = _capi.load()
    rng = np.random.default_rng(1)
    for _ in range(200):
        x0 = _state(t=rng.uniform(0, 5), s=rng.uniform(0, 200), v=rng.uniform(0, 36), a=rng.uniform(-3, 1.5))
        j, dt = float(rng.uniform(-3, 2)), float(rng.uniform(0.05, 2))
        a, b = TplLonState(), TplLonState()
        O.lib().dpo_lon_dynamics_host(C.byref(x0), j, dt, C.byref(a))
        lib.tplhip_lon_dynamics(C.byref(x0), j, dt, C.byref(b))
        assert bytes(a) == bytes(b)
    traj = tpl_amd.LonTraj()
    for i in range(5):
        st = tpl_amd.LonState()
        st.t, st.s, st.v, st.a, st.j = float(i), 8.0 * i, 8.0, 0.1, 0.2
        traj.states.append(st)
    arr = traj._as_array()
    for t in [-0.5, 0.0, 0.3, 2.0, 3.999, 4.0, 7.0]:
        a = TplLonState()
        O.lib().dpo_traj_state(arr, 5, t, C.byref(a))
        b = traj.state(t)
        assert bytes(a) == bytes(b._c)
    many = traj.state(np.array([0.3, 2.0]))
    assert many.shape == (2, 4)


# ---- rule R1 = 1 (explicit contraction, the default): its arithmetic pinned WITHOUT a libm or a compiler in the
# loop -- fused multiply-adds emulated in exact rational arithmetic with one round-to-nearest-even at the end
def _rn(x, p, emin):
    """Round the exact rational x to a binary float with p significand bits (ties to even; subnormals below 2^emin)."""
    from fractions import Fraction
    if x == 0:
        return Fraction(0)
    sign = -1 if x < 0 else 1
    x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if Fraction(2) ** e > x:
        e -= 1
    e = max(e, emin)
    q = Fraction(2) ** (e - p + 1)
    n = x / q
    lo = n.numerator // n.denominator
    rem = n - lo
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and lo % 2 == 1):
        lo += 1
    return sign * lo * q


def _f32(x):
    return _rn(x, 24, -126)


def _f64(x):
    return _rn(x, 53, -1022)


def test_explicit_contraction_of_lon_dynamics_and_interp_path_is_the_documented_one():
    """lonDynamics (lon_planner.cu:240-250) and interpPath (common.cuh:124-136) under R1 = 1 (the default) against the rule
    written in the header of oracle/dp_lon_oracle.c -- fmaf(v, dt, s) -> fma((.5 a) dt, dt, .) -> fma(((j / 6) dt) dt, dt, .);
    fmaf(a, dt, v) -> fma((.5 j) dt, dt, .); fmaf(j, dt, a); fmaf(p0, 1 - alpha, p1 alpha) -- evaluated in exact
    rationals, one rounding per (fused) operation.  And R1 = 0 is the same rule with every fma split in two."""
    from fractions import Fraction as F
    rng = np.random.default_rng(11)
    for fma in (1, 0):
        O.set_rule(fma=fma)
        try:
            for _ in range(200):
                s, v, a, j, dt = (np.float32(x) for x in (rng.uniform(0, 150), rng.uniform(0, 30), rng.uniform(-3, 1.5),
                                                           rng.uniform(-3, 2), rng.choice([0.3, 0.71, 1.0, 1.25])))
                x0 = _state(s=s, v=v, a=a)
                out = TplLonState()
                O.lib().dpo_lon_dynamics(C.byref(x0), j, dt, C.byref(out))
                S, V, A, J, DT = (F(float(x)) for x in (s, v, a, j, dt))
                sixth = F(1.0 / 6.0)
                if fma:
                    sf = _f32(V * DT + S)
                    sd = _f64(_f64(_f64(F(1, 2) * A) * DT) * DT + sf)
                    sd = _f64(_f64(_f64(_f64(sixth * J) * DT) * DT) * DT + sd)
                    vf = _f32(A * DT + V)
                    vd = _f64(_f64(_f64(F(1, 2) * J) * DT) * DT + vf)
                    an = _f32(J * DT + A)
                else:
                    sf = _f32(_f32(V * DT) + S)
                    sd = _f64(_f64(_f64(_f64(F(1, 2) * A) * DT) * DT) + sf)
                    sd = _f64(_f64(_f64(_f64(_f64(sixth * J) * DT) * DT) * DT) + sd)
                    vf = _f32(_f32(A * DT) + V)
                    vd = _f64(_f64(_f64(_f64(F(1, 2) * J) * DT) * DT) + vf)
                    an = _f32(_f32(J * DT) + A)
                assert F(float(out.s)) == max(S, _f32(sd)), (fma, s, v, a, j, dt)
                assert F(float(out.v)) == max(F(0), _f32(vd)), (fma, s, v, a, j, dt)
                assert F(float(out.a)) == an, (fma, s, v, a, j, dt)
            path = sc.straight_path(50, 0.7)
            path[:, 5] = rng.uniform(5, 15, 50).astype(np.float32)
            path[:, 2] = (path[:, 6] * np.float32(0.97)).astype(np.float32)
            for _ in range(100):
                d = np.float32(rng.uniform(0.0, 33.0))
                ps = TplPathState()
                O.lib().dpo_interp_path(O.fptr(path), d, np.float32(0.7), 50, C.byref(ps))
                alpha = _f32(F(float(d)) / F(float(np.float32(0.7))))
                i0 = min(49, max(0, int(alpha.numerator // alpha.denominator)))
                i1 = min(49, max(0, -int((-alpha.numerator) // alpha.denominator)))
                al = _f32(alpha - i0)
                ia = _f32(_f64(1 - al))
                for col, got in ((5, ps.v_max), (2, ps.s)):
                    p0, p1 = F(float(path[i0, col])), F(float(path[i1, col]))
                    want = _f32(p0 * ia + _f32(p1 * al)) if fma else _f32(_f32(p0 * ia) + _f32(p1 * al))
                    assert F(float(got)) == want, (fma, d, col)
        finally:
            O.set_rule()


@pytest.mark.parametrize("name,order", [(None, 0), ("order_sva", 1), ("sum8_f32", 2), ("sum8_wide", 3)])
def test_read_node_blend_orders_are_the_documented_ones(name, order):
    """readNode (lon_planner.cu:26-38 through the software filter of rules R2 / R3) under the default order and the three
    other blend orders (dpo_set_rule(., ., ., 1 | 2 | 3)) against the rule written in the header of oracle/dp_lon_oracle.c,
    evaluated in exact rationals with one rounding per operation: three sequential blends fma(q, w, p (1 - w)) along
    a, v, s (default) or s, v, a (order 1); the eight-corner sum with weights (w_s w_v) w_a, corner index s + 2 v + 4 a,
    products and sums rounded in fp32 in that order (order 2) or exact products in a double accumulator and one final
    rounding (order 3).  Pins the corner indexing and the operation order of every variant independently of the C."""
    from fractions import Fraction as F
    p, _ = _small()
    rng = np.random.default_rng(21 + order)
    table = rng.uniform(0, 10, size=(p.a_steps, p.v_steps, p.s_steps, 4)).astype(np.float32)
    out = np.zeros(4, dtype=np.float32)

    def tap(x, x_min, x_max, n):
        t1 = _f32(F(float(x)) - F(float(x_min)))
        t2 = _f32(F(float(x_max)) - F(float(x_min)))
        c = _f32(_f32(t1 / t2) * (n - 1))
        xb = _f32(_f32(c + F(1, 2)) - F(1, 2))
        fl = xb.numerator // xb.denominator
        alpha = xb - fl
        q = alpha * 256
        r = q.numerator // q.denominator                      # rint: to nearest, ties to even
        rem = q - r
        if rem > F(1, 2) or (rem == F(1, 2) and r % 2 == 1):
            r += 1
        return min(n - 1, max(0, fl)), min(n - 1, max(0, fl + 1)), F(r, 256)

    def lerp(pv, qv, w):
        return _f32(qv * w + _f32(pv * _f32(1 - w)))            # fmaf(q, w, p * (1 - w))

    O.set_rule(order=order)
    try:
        for _ in range(300):
            s = np.float32(rng.uniform(p.s_min, p.s_max))
            v = np.float32(rng.uniform(p.v_min, p.v_max))
            a = np.float32(rng.uniform(p.a_min, p.a_max))
            O.lib().dpo_read_node(C.byref(p), O.fptr(table), s, v, a, O.fptr(out))
            s0, s1, ws = tap(s, p.s_min, p.s_max, p.s_steps)
            v0, v1, wv = tap(v, p.v_min, p.v_max, p.v_steps)
            a0, a1, wa = tap(a, p.a_min, p.a_max, p.a_steps)
            for ch in range(3):
                T = lambda ia, iv, i_s: F(float(table[ia, iv, i_s, ch]))
                if order == 0:
                    lo = lerp(lerp(T(a0, v0, s0), T(a1, v0, s0), wa), lerp(T(a0, v1, s0), T(a1, v1, s0), wa), wv)
                    hi = lerp(lerp(T(a0, v0, s1), T(a1, v0, s1), wa), lerp(T(a0, v1, s1), T(a1, v1, s1), wa), wv)
                    want = lerp(lo, hi, ws)
                elif order == 1:
                    y0 = lerp(lerp(T(a0, v0, s0), T(a0, v0, s1), ws), lerp(T(a0, v1, s0), T(a0, v1, s1), ws), wv)
                    y1 = lerp(lerp(T(a1, v0, s0), T(a1, v0, s1), ws), lerp(T(a1, v1, s0), T(a1, v1, s1), ws), wv)
                    want = lerp(y0, y1, wa)
                else:
                    s2, v2, a2 = (_f32(1 - ws), ws), (_f32(1 - wv), wv), (_f32(1 - wa), wa)
                    corners = [(i_s, iv, ia) for ia in (0, 1) for iv in (0, 1) for i_s in (0, 1)]       # index s + 2 v + 4 a
                    w = [_f32(_f32(s2[i_s] * v2[iv]) * a2[ia]) for i_s, iv, ia in corners]
                    t = [T((a0, a1)[ia], (v0, v1)[iv], (s0, s1)[i_s]) for i_s, iv, ia in corners]
                    if order == 2:
                        acc = _f32(w[0] * t[0])
                        for c in range(1, 8):
                            acc = _f32(acc + _f32(w[c] * t[c]))
                        want = acc
                    else:
                        acc = _f64(w[0] * t[0])
                        for c in range(1, 8):
                            acc = _f64(acc + _f64(w[c] * t[c]))
                        want = _f32(acc)
                assert F(float(out[ch])) == want, (order, ch, float(s), float(v), float(a), float(out[ch]), float(want))
    finally:
        O.set_rule()


@pytest.mark.parametrize("name,order", [(None, 0), ("order_sva", 1), ("sum8_f32", 2), ("sum8_wide", 3)])
def test_dist_map_lookup_under_the_blend_orders(name, order):
    """interpDistMapPath (env.cu:254-264: x = t, y = s, z = 0) at the integral t index the planner asks for: every sequential
    order gives the blend along s alone (a weight of exactly 0 leaves p for finite q), the sums the two non-zero corner terms
    (1 - w) p + w q, rounded per product and sum in fp32 (order 2) or once (order 3)."""
    from fractions import Fraction as F
    p, ep = _small()
    rng = np.random.default_rng(5)
    dist = rng.uniform(0.0, 120.0, size=(ep.t_steps, ep.s_steps)).astype(np.float32)
    O.set_rule(order=order)
    try:
        lib = O.lib()
        for _ in range(300):
            it = int(rng.integers(0, ep.t_steps))
            t = np.float32(0.0) if it == 0 else np.float32(ep.dt_start + (it - 1) * ep.dt)
            s = np.float32(rng.uniform(ep.s_min, ep.s_max))
            got = lib.dpo_interp_dist_map_path(C.byref(ep), O.fptr(dist), t, s)
            t1 = _f32(F(float(s)) - F(float(ep.s_min)))
            t2 = _f32(F(float(ep.s_max)) - F(float(ep.s_min)))
            c = _f32(_f32(t1 / t2) * (ep.s_steps - 1))
            xb = _f32(_f32(c + F(1, 2)) - F(1, 2))
            fl = xb.numerator // xb.denominator
            q = (xb - fl) * 256
            r = q.numerator // q.denominator
            rem = q - r
            if rem > F(1, 2) or (rem == F(1, 2) and r % 2 == 1):
                r += 1
            w = F(r, 256)
            i0, i1 = min(ep.s_steps - 1, max(0, fl)), min(ep.s_steps - 1, max(0, fl + 1))
            pv, qv = F(float(dist[it, i0])), F(float(dist[it, i1]))
            if order <= 1:
                want = _f32(qv * w + _f32(pv * _f32(1 - w)))
            elif order == 2:
                want = _f32(_f32(_f32(1 - w) * pv) + _f32(w * qv))        # (the two zero-weight corners add +0)
            else:
                want = _f32(_f64(_f64(_f32(1 - w) * pv) + _f64(w * qv)))
            assert F(float(got)) == want, (order, it, float(s), got, float(want))
    finally:
        O.set_rule()


def test_rule_is_run_time_state_with_the_contracted_default():
    """ONE checker library, the arithmetic rule a setting of it (round 6; until then one build per rule): the default is the
    contracted rule {rint, fma form, R1 = 1, a -> v -> s}, `O.rule(...)` restores it on exit -- also when the body raises --,
    values outside the rule's ranges are refused, and the rule really reaches the arithmetic (R1 moves a grid point)."""
    assert O.get_rule() == O.DEFAULT_RULE == dict(quant=1, lerp=0, fma=1, order=0)
    with O.rule("wfloor", fma=0, order=2) as r:
        assert r == dict(quant=2, lerp=0, fma=0, order=2) == O.get_rule()
    assert O.get_rule() == O.DEFAULT_RULE
    with pytest.raises(ZeroDivisionError):
        with O.rule(order=3):
            1 / 0
    assert O.get_rule() == O.DEFAULT_RULE
    for bad in (dict(quant=3), dict(lerp=-1), dict(fma=2), dict(order=4)):
        with pytest.raises(AssertionError):
            O.set_rule(**bad)
        assert O.get_rule() == O.DEFAULT_RULE                # a refused rule changes nothing
    # fma(i, step, min) against RN(RN(i * step) + min): some grid point of the a-axis of configs[1] differs in its last bit
    step = np.float32(4.5) / np.float32(31)
    fused = [O.grid_point(-3.0, i, step) for i in range(32)]
    with O.rule(fma=0):
        plain = [O.grid_point(-3.0, i, step) for i in range(32)]
    assert fused != plain and max(abs(a - b) for a, b in zip(fused, plain)) < 5e-7
