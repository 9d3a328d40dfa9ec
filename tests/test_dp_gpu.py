"""GPU parity tests of HP-1: the HIP planner (through the C-ABI of libtplhip.so)
against the CPU oracle on identical inputs.  Bar: bit-exact -- value tables
(cost, constr, argmin jerk), distance maps and trajectories compare as raw
bytes, which is stronger than "argmin indices bit-exact"."""
import ctypes as C

import numpy as np
import pytest

import oracle_dp as O
import tpl_amd
from benchkit import scenarios as sc
from tpl_amd._capi import TplLonState

pytestmark = pytest.mark.gpu


def _traj_np(traj):
    return O.states_to_np([s._c for s in traj.states])


def _run_both(p, ep, path, occ, inits):
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    trajs = dp.update_batch(inits, path, env)
    cp, cep = O.cparams(p), O.cparams(ep)
    tables, otrajs, dist = O.update(cp, cep, path, occ, inits)
    return dp, env, trajs, tables, otrajs, dist


def _assert_tables_equal(dp, tables, T):
    for t in range(1, T):
        got = dp.copy_table(t)
        want = tables[t]
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
            i = tuple(bad[0])
            raise AssertionError(f"stage {t}: {len(bad)} words differ, first at {i}: got {got[i]!r} want {want[i]!r}")


GRID_CASES = [
    # s, v, a, t, obstacles, dt_start
    pytest.param(33, 9, 7, 5, sc.OBSTACLES_OVERTAKE, None, id="tiny-33x9x7"),
    pytest.param(64, 12, 9, 6, sc.random_obstacles(4, 8), None, id="s64-exact-wave"),
    pytest.param(201, 37, 20, 10, sc.OBSTACLES_OVERTAKE_TRAFFIC, None, id="reference-default-201x37x20"),
    pytest.param(201, 37, 20, 10, sc.OBSTACLES_OVERTAKE, 0.6, id="default-dt_start-0.6"),
    pytest.param(96, 10, 8, 6, (), 0.3, id="dt_start-0.3-rule-R4"),
    pytest.param(512, 8, 6, 4, sc.random_obstacles(7, 5), None, id="s512-kchunk3"),
    pytest.param(1000, 6, 5, 3, sc.random_obstacles(8, 5), None, id="s1000-kchunk1"),
    # grids several residency rounds deep select the two-rows-per-lane sweep (k_backward3v):
    pytest.param(201, 48, 44, 4, sc.OBSTACLES_OVERTAKE_TRAFFIC, None, id="ragged-rows-v48"),
    pytest.param(201, 50, 42, 4, sc.random_obstacles(3, 6), 0.4, id="ragged-rows-v50-dt_start"),
    pytest.param(256, 40, 52, 4, sc.OBSTACLES_OVERTAKE, None, id="full-rows-v40"),
    pytest.param(128, 64, 66, 3, sc.random_obstacles(9, 7), None, id="s128-v64"),
    # an odd number of velocity rows: the last row pair of k_backward3v is row V-1 twice
    pytest.param(128, 37, 120, 3, sc.random_obstacles(5, 6), None, id="odd-v-steps-plain-fp32-sweep"),
    # 64 acceleration rows: the jerk grid lands on the acceleration grid (blend weight exactly 0 for three of the nine
    # actions) next to the clamped ones -- both kinds of ACT_SINGLE_A, in both plain-fp32 sweeps
    pytest.param(128, 64, 64, 3, sc.OBSTACLES_OVERTAKE_TRAFFIC, None, id="a64-weight-zero-blends-3v"),
    pytest.param(96, 10, 64, 4, sc.random_obstacles(6, 9), 0.4, id="a64-weight-zero-blends-3s"),
]

EXPECTED_KERNEL = {"ragged-rows-v48": "k_backward3v", "ragged-rows-v50-dt_start": "k_backward3v", "full-rows-v40": "k_backward3v",
                   "s128-v64": "k_backward3v", "odd-v-steps-plain-fp32-sweep": "k_backward3v",
                   "reference-default-201x37x20": "k_backward3s", "tiny-33x9x7": "k_backward3s",
                   "s512-kchunk3": "k_backward3v", "s1000-kchunk1": "k_backward3v",     # rows wider than 256 lanes on a small grid: the two-rows form
                   "a64-weight-zero-blends-3v": "k_backward3v", "a64-weight-zero-blends-3s": "k_backward3s"}


@pytest.mark.parametrize("S,V,A,T,obst,dt_start", GRID_CASES)
def test_tables_and_trajectories_bit_exact(S, V, A, T, obst, dt_start, request):
    p = sc.lon_params(S, V, A, T, dt_start=dt_start)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31, dt_start=dt_start)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, obst)
    inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(6, seed=5)
    dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
    assert np.array_equal(env.get_dist_map_path().view(np.uint32), dist.view(np.uint32))
    case = request.node.callspec.id
    if case in EXPECTED_KERNEL:
        info = dp.info()
        assert info["sweep"] == EXPECTED_KERNEL[case], info    # bounded inputs, power-of-two path step: the plain-fp32 sweeps
    _assert_tables_equal(dp, tables, T)
    for r, tr in enumerate(trajs):
        got = _traj_np(tr)
        assert np.array_equal(got.view(np.uint32), otrajs[r].view(np.uint32)), (r, got, otrajs[r])


@pytest.mark.parametrize("S,V,A,T,order", [(128, 64, 66, 3, 0), (128, 37, 120, 3, 0), (201, 50, 42, 4, 0), (96, 21, 30, 3, 0),
                                             (128, 37, 120, 3, 1), (201, 50, 42, 3, 2)])
def test_skewed_traversal_writes_every_row(S, V, A, T, order, monkeypatch):
    """The skewed traversal of the two-rows-per-lane sweeps (default on tables beyond ~100 MB: config 4) is a per-a-row
    rotation of row pairs (dp_sweep.hip: `lp -= (ia * skew) % vx`); a wrong modulus leaves a row pair unwritten or
    written twice.  TPLHIP_DEBUG=skew=1,big=1 forces it (and the large-grid kernel) on grids the oracle sweeps whole:
    FULL tables compared, incl. an odd number of velocity rows (last pair = row V-1 twice), XCD shares of unequal size,
    and the ordered form k_backward3o which shares the traversal."""
    monkeypatch.setenv("TPLHIP_DEBUG", "skew=1,big=1")
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.random_obstacles(5, 6))
    inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(2, seed=5)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    if order:
        dp.set_blend_order(order)
    trajs = dp.update_batch(inits, path, env)
    with O.rule(order=order):
        tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
    info = dp.info()
    assert info["sweep"] == ("k_backward3o" if order else "k_backward3v") and info["skew_pairs_per_a_row"] > 0.0, info
    _assert_tables_equal(dp, tables, T)
    for r, tr in enumerate(trajs):
        assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), r


@pytest.mark.parametrize("S,V,A,T,dt_start", [(201, 37, 20, 10, None), (201, 37, 20, 6, 0.6), (128, 64, 32, 5, None),
                                              (70, 9, 7, 4, 0.3), (256, 12, 6, 3, None), (33, 3, 2, 3, None)])
def test_small_grid_forms_agree_bit_for_bit(S, V, A, T, dt_start, monkeypatch):
    """Grids that leave the chip mostly empty have two sweeps: k_backward (scalar-operand instruction stream, two-pass
    argmin; TPLHIP_DEBUG=no_s3) and k_backward3s (plain-fp32 stream, integer-key argmin; the default).  Each against the
    oracle, tables and trajectories as raw bits."""
    p = sc.lon_params(S, V, A, T, dt_start=dt_start)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31, dt_start=dt_start)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
    inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(3, seed=11)
    for form in ("k_backward3s", "k_backward"):
        monkeypatch.setenv("TPLHIP_DEBUG", "no_s3" if form == "k_backward" else "")
        dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
        info = dp.info()
        assert info["big"] == 0 and info["sweep"] == form, info
        _assert_tables_equal(dp, tables, T)
        for r, tr in enumerate(trajs):
            assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (form, r)


def test_fuzz_sample():
    """A short sample of the randomised soak (tests/fuzz_dp.py; the long run is a tool, not a test)."""
    import fuzz_dp
    for seed in range(12):
        fuzz_dp.run_case(seed)
    for seed in range(100, 112):               # ... and under randomly drawn arithmetic rules (R2 x R3 x R1)
        fuzz_dp.run_case(seed, rules=True)
    for seed in range(200, 212):               # ... and under a random blend order on top (R3's order / form)
        fuzz_dp.run_case(seed, rules=True, orders=True)


def test_single_row_axes_are_refused():
    """One sample on an axis makes the reference's step size (max - min) / (steps - 1) a division by zero; here the
    planner says so instead of planning on NaNs."""
    dp = tpl_amd.DynProgLonPlanner()
    for S, V, A in ((3, 17, 1), (9, 1, 5), (1, 4, 4)):
        with pytest.raises(ValueError, match="s/v/a_steps >= 2"):
            dp.reinit_buffers(sc.lon_params(S, V, A, 4))


@pytest.mark.parametrize("S,V,A,T", [(2, 2, 2, 2), (5, 2, 3, 2), (64, 2, 2, 3), (70, 3, 2, 2)])
def test_degenerate_grids_still_match(S, V, A, T):
    """The smallest grids the parameters allow (two stations, two stages, two speeds, two accelerations): same
    arithmetic, same answers -- and no launch geometry that divides by zero or reads past a row."""
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    inits = [sc.initial_state(0.0, 8.0, 0.0), sc.initial_state(20.0, 2.0, -1.0)]
    dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
    _assert_tables_equal(dp, tables, T)
    for r, tr in enumerate(trajs):
        got = _traj_np(tr)
        assert np.array_equal(got.view(np.uint32), otrajs[r].view(np.uint32)), (r, got, otrajs[r])


def test_config2_full_plan_bit_exact():
    # BASELINE configs[1]: 128x64x32 grid, 50 stages (oracle: ~1e8 action evaluations)
    S, V, A, T = sc.GRIDS["config2"]
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    inits = [sc.initial_state(0.0, 8.0, 0.0)]
    dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
    _assert_tables_equal(dp, tables, T)
    assert np.array_equal(_traj_np(trajs[0]).view(np.uint32), otrajs[0].view(np.uint32))
    # single-rollout entry point returns the same plan
    single = dp.update(inits[0], path, env)
    assert np.array_equal(_traj_np(single).view(np.uint32), otrajs[0].view(np.uint32))


def test_reeval_matches_oracle_and_checks_length():
    S, V, A, T = 65, 10, 8, 6
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=101, l_steps=21)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    init = sc.initial_state(0.0, 9.0, 0.2)
    dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, [init])
    # perturb the stored jerks, re-cost on both sides
    traj = trajs[0].copy()
    for i, st in enumerate(traj.states):
        st.j = st.j + 0.1 * ((i % 3) - 1)
    got = dp.reeval_traj(traj, path, env)
    cp, cep = O.cparams(p), O.cparams(ep)
    arr = traj._as_array()
    out = (TplLonState * T)()
    rc = O.lib().dpo_reeval(C.byref(cp), C.byref(cep), O.fptr(path), O.fptr(dist), O.fptr(tables),
                            arr, T, out)
    assert rc == 0
    assert np.array_equal(_traj_np(got).view(np.uint32), O.states_to_np(out).view(np.uint32))
    # lon_planner.cu:423-425: wrong length -> RuntimeError
    short = tpl_amd.LonTraj()
    short.states = traj.states[:-1]
    with pytest.raises(RuntimeError, match="trajectory size does not match t_steps"):
        dp.reeval_traj(short, path, env)


def test_error_behaviour():
    p = sc.lon_params(33, 9, 7, 5)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    env = tpl_amd.DynProgEnvironment()
    with pytest.raises(ValueError, match="incorrect shape"):
        dp.update(sc.initial_state(), np.zeros((10, 7), dtype=np.float32), env)
    bad = sc.lon_params(2000, 9, 7, 5)
    with pytest.raises(ValueError, match="s_steps"):
        dp.reinit_buffers(bad)


def test_reinit_keeps_tables_when_only_weights_change():
    p = sc.lon_params(64, 10, 8, 5)
    ep = sc.env_params(t_steps=5, s_steps=101, l_steps=21)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, ())
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    a = dp.update(sc.initial_state(), path, env)
    p.dt_start = 0.7           # what the Python caller changes every cycle
    ep.dt_start = 0.7
    env.reinit_buffers(ep)
    dp.reinit_buffers(p)
    b = dp.update(sc.initial_state(), path, env)
    cp, cep = O.cparams(p), O.cparams(ep)
    _, otrajs, _ = O.update(cp, cep, path, occ, [sc.initial_state()])
    assert np.array_equal(_traj_np(b).view(np.uint32), otrajs[0].view(np.uint32))
    assert not np.array_equal(_traj_np(a), _traj_np(b))


@pytest.mark.parametrize("grid", ["target", "config4"])
def test_full_size_spot_checks(grid):
    """BASELINE full sizes: the oracle cannot sweep these in seconds, so check
    size-independent properties: (1) the terminal stage against its closed form,
    (2) random cells of two stages re-evaluated by the oracle's evalNode on the
    GPU's own next-stage table (bit-exact), (3) the trajectory re-costed by
    reeval equals the forward pass, stations are monotone, t advances by dt."""
    S, V, A, T = sc.GRIDS[grid]            # config4: 512x256x128 x 100 stages = 20 GB of value tables, as stated
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    init = sc.initial_state(0.0, 8.0, 0.0)
    traj = dp.update(init, path, env)
    cp, cep = O.cparams(p), O.cparams(ep)
    dist = O.dist_map_path(cep, occ, path, cp.path_steps, cp.path_step_size)
    assert np.array_equal(env.get_dist_map_path().view(np.uint32), dist.view(np.uint32))

    rng = np.random.default_rng(11)
    for t in (T - 1, T - 2, T // 2, 1):
        nxt = dp.copy_table(min(T - 1, t + 1))
        cur = dp.copy_table(t)
        n = 1000
        ia = rng.integers(0, A, n); iv = rng.integers(0, V, n); i_s = rng.integers(0, S, n)
        for k in range(n):
            tp = TplLonState()
            tp.t = O.grid_point(cp.dt_start, t - 1, cp.dt)
            tp.s = O.grid_point(cp.s_min, i_s[k], cp._s_step_size)
            tp.v = O.grid_point(cp.v_min, iv[k], cp._v_step_size)
            tp.a = O.grid_point(cp.a_min, ia[k], cp._a_step_size)
            O.lib().dpo_eval_node(C.byref(cp), C.byref(cep), 9, C.byref(tp), O.fptr(nxt), O.fptr(path),
                                  O.fptr(dist), cp.dt, 1, 1 if t == T - 1 else 0)
            want = np.array([tp.cost, tp.constr, tp.j, 0.0], dtype=np.float32)
            got = cur[ia[k], iv[k], i_s[k]]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (t, ia[k], iv[k], i_s[k], got, want)

    # ... and one whole random station column -- EVERY (v, a) row -- of two stages: the traversal of the large-grid sweep
    # (XCD shares, strips, the skewed rotation of row pairs on tables beyond ~100 MB) decides which workgroup writes
    # which row pair; random cells find one unwritten pair of 16 384 with p ~ 0.2, a column finds it always
    for t in (T - 2, 2):
        nxt = dp.copy_table(min(T - 1, t + 1))
        cur = dp.copy_table(t)
        i_s = int(rng.integers(0, S))
        for ia in range(A):
            for iv in range(V):
                tp = TplLonState()
                tp.t = O.grid_point(cp.dt_start, t - 1, cp.dt)
                tp.s = O.grid_point(cp.s_min, i_s, cp._s_step_size)
                tp.v = O.grid_point(cp.v_min, iv, cp._v_step_size)
                tp.a = O.grid_point(cp.a_min, ia, cp._a_step_size)
                O.lib().dpo_eval_node(C.byref(cp), C.byref(cep), 9, C.byref(tp), O.fptr(nxt), O.fptr(path),
                                      O.fptr(dist), cp.dt, 1, 0)
                want = np.array([tp.cost, tp.constr, tp.j, 0.0], dtype=np.float32)
                got = cur[ia, iv, i_s]
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), ("column", t, ia, iv, i_s, got, want)

    tr = _traj_np(traj)
    assert np.all(np.diff(tr[:, 1]) >= 0.0)
    assert np.allclose(np.diff(tr[:, 0]), cp.dt)
    again = _traj_np(dp.reeval_traj(traj, path, env))
    assert np.array_equal(again[:, 5:].view(np.uint32), tr[:, 5:].view(np.uint32))


@pytest.mark.parametrize("S,V,A,T,kernel", [(201, 20, 12, 6, "k_backward3s"), (128, 64, 66, 3, "k_backward3v")])
def test_division_modes_agree(S, V, A, T, kernel, monkeypatch):
    """The plain-fp32 sweeps form their two divisions by run-time constants in the cheapest form reinit proved exact by
    brute force (k_verify_div): one multiplication (power-of-two divisor), a two-float reciprocal, a reciprocal with
    one Newton step on the quotient -- or as IEEE divisions where no proof holds.  Every form on the same kernel, and
    the general sweep: the same bits."""
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    tabs = {}
    forms = {}
    for mode, dbg in (("fast", ""), ("ieee", "no_fastdiv"), ("form3", "div_form=3"), ("form5", "div_form=5"),
                      ("general", "no_v3,no_s3")):
        monkeypatch.setenv("TPLHIP_DEBUG", dbg)
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        env.set_occ_map(occ)
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p)
        dp.update(sc.initial_state(0.0, 8.0, 0.0), path, env)
        info = dp.info()
        assert info["sweep"] == ("k_backward" if mode == "general" else kernel), info
        forms[mode] = (info["path_div3"], info["range_div3"])
        tabs[mode] = [dp.copy_table(t) for t in range(1, T)]
    assert forms["ieee"] == (0, 0) and forms["fast"] == (4, 3), forms         # path step 1 m, range 200 m
    assert forms["form5"] == (5, 5) and forms["form3"][1] == 3, forms
    for mode in tabs:
        for a, b in zip(tabs["fast"], tabs[mode]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), mode
    monkeypatch.setenv("TPLHIP_DEBUG", "")
    # divisors that used to send the grid to the general sweep (0.3 / 0.7 / 1.3 m path steps, a 157.3 m range): they stay
    # on the plain-fp32 sweep, in whatever form holds for them, and match the oracle
    for step, s_max in ((0.3, 200.0), (0.7, 157.3), (1.3, 157.3)):
        p2 = sc.lon_params(S, V, A, T)
        p2.path_step_size = step
        p2.path_steps = int(np.ceil(260.0 / step))
        p2.s_max = s_max
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p2)
        info = dp.info()
        assert info["path_div3"] in (0, 3, 4, 5) and info["range_div3"] in (0, 3, 4, 5), info
        path2 = sc.straight_path(p2.path_steps, p2.path_step_size)
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        env.set_occ_map(occ)
        trajs = dp.update_batch([sc.initial_state(0.0, 8.0, 0.0)], path2, env)
        assert dp.info()["sweep"] == kernel, dp.info()
        cp, cep = O.cparams(p2), O.cparams(ep)
        tables, otrajs, _ = O.update(cp, cep, path2, occ, [sc.initial_state(0.0, 8.0, 0.0)])
        _assert_tables_equal(dp, tables, T)
        assert np.array_equal(_traj_np(trajs[0]).view(np.uint32), otrajs[0].view(np.uint32))


def test_pipelined_downloads_match_the_blocking_one():
    """stage_download / collect_download (two slots, a plan enqueued behind each copy) hand out the
    same trajectories as the blocking download of the same plans."""
    S, V, A, T = 64, 24, 12, 8
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(sc.occ_map(ep, sc.OBSTACLES_OVERTAKE))
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    inits = [[sc.initial_state(0.0, 4.0 + k, 0.0)] for k in range(4)]
    want = []
    for ini in inits:
        dp.upload(ini, path)
        dp.plan_resident(env)
        want.append(_traj_np(dp.download()[0]))
    got = []
    for k, ini in enumerate(inits):
        dp.upload(ini, path)
        dp.plan_resident(env)
        dp.stage_download(k & 1)
        if k > 0:
            got.append(_traj_np(dp.collect_download((k - 1) & 1)[0]))
    got.append(_traj_np(dp.collect_download((len(inits) - 1) & 1)[0]))
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    with pytest.raises(RuntimeError, match="no download staged"):
        dp.collect_download(0)


@pytest.mark.parametrize("S,V,A,T,cycles", [(201, 37, 20, 6, 10), (128, 64, 66, 4, 4)], ids=["small-grid-sweep", "two-rows-per-lane-sweep"])
def test_one_wait_per_plan_keeps_every_plan_exact(S, V, A, T, cycles):
    """update() waits for the device once per plan: a re-parameterisation (dt_start moves every cycle in tpl) only
    enqueues the new action table, path and start state go through a page-locked staging buffer without a wait.
    Cycles with a changing dt_start (alone among the parameters: the action table is kept), path and start state, each
    against the oracle, on a grid of each sweep family; then the staging buffer
    reused before anything waited (upload twice), and a download repeated."""
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
    env = tpl_amd.DynProgEnvironment()
    dp = tpl_amd.DynProgLonPlanner()
    p = sc.lon_params(S, V, A, T)
    base = sc.straight_path(p.path_steps, p.path_step_size)
    for k in range(cycles):
        p.dt_start = 1.0 - 0.07 * k
        ep.dt_start = p.dt_start
        path = base.copy()
        path[:, 5] = 13.9 - 0.3 * k                          # speed limit of the cycle
        init = sc.initial_state(0.5 * k, 6.0 + 0.4 * k, 0.1 * (k % 3))
        env.reinit_buffers(ep)
        env.set_occ_map(occ)
        dp.reinit_buffers(p)
        got = _traj_np(dp.update(init, path, env))
        tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, [init])
        assert np.array_equal(got.view(np.uint32), otrajs[0].view(np.uint32)), k
        if k in (0, cycles - 1):
            _assert_tables_equal(dp, tables, T)
    other = sc.initial_state(3.0, 11.0, -0.5)
    dp.upload([init], path)
    dp.upload([other], path)                                 # the staging buffer of the first upload is still in flight
    dp.plan_resident(env)
    first = _traj_np(dp.download()[0])
    again = _traj_np(dp.download()[0])
    _, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, [other])
    assert np.array_equal(first.view(np.uint32), otrajs[0].view(np.uint32))
    assert np.array_equal(again.view(np.uint32), first.view(np.uint32))


@pytest.mark.parametrize("S,V,A,T", [(201, 37, 20, 10), (128, 64, 66, 6), (256, 128, 64, 4)],
                         ids=["small-grid-sweep", "two-rows-per-lane-sweep", "headline-grid"])
def test_plans_in_flight_on_two_planners_equal_the_oracle(S, V, A, T):
    """bench.py's headline deals its plans to TWO planners (own stream, tables and environment each) from one host
    thread, launches interleaved and nothing waited for in between: the sweeps of both plans share the chip.  Different
    scenario per planner, six plans each with a new start state every time -- every trajectory and the last plan's
    tables equal the oracle's, whichever planner's workgroups ran beside them."""
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occs = [sc.occ_map(ep, sc.OBSTACLES_OVERTAKE), sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)]
    fleet = []
    for occ in occs:
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        env.set_occ_map(occ)
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p)
        fleet.append((dp, env))
    n = 6 if S * V * A < 1 << 20 else 2
    inits = [[sc.initial_state(0.3 * k + q, 5.0 + 0.7 * k + q, 0.1 * ((k + q) % 3)) for q in range(2)] for k in range(n)]
    got = [[], []]
    for k in range(n):
        for q, (dp, env) in enumerate(fleet):
            if k > 0:
                got[q].append(_traj_np(dp.collect_download((k - 1) & 1)[0]))
            dp.upload([inits[k][q]], path)
            dp.plan_resident(env)
            dp.stage_download(k & 1)
    for q, (dp, env) in enumerate(fleet):
        got[q].append(_traj_np(dp.collect_download((n - 1) & 1)[0]))
    for q, (dp, env) in enumerate(fleet):
        for k in range(n):
            tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occs[q], [inits[k][q]])
            assert np.array_equal(got[q][k].view(np.uint32), otrajs[0].view(np.uint32)), (q, k)
        _assert_tables_equal(dp, tables, T)


@pytest.mark.parametrize("S,V,A,T", [(201, 37, 20, 10), (128, 64, 32, 12)])
def test_replanning_with_new_path_start_and_environment_back_to_back(S, V, A, T):
    """What a caller that replans every cycle does, without waiting for anything in between: upload(new start, NEW PATH),
    plan_resident(ANOTHER environment), stage_download -- eight plans on ONE planner.  In the back-to-back mode the rollout of
    plan k runs on its own stream while upload k + 1 overwrites the planner's path and start state and plan k + 1's front
    rewrites a distance map: the rollout reads its own copies (round 6: no host wait in upload), so every plan must equal
    the oracle's plan of ITS inputs -- start state, speed-limit profile of its path, obstacle set."""
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    obst = [sc.OBSTACLES_OVERTAKE, sc.OBSTACLES_OVERTAKE_TRAFFIC, sc.random_obstacles(4, 8), ()]
    occs = [sc.occ_map(ep, o) for o in obst]
    envs = []
    for occ in occs:
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        env.set_occ_map(occ)
        envs.append(env)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    n = 8
    jobs = []
    for k in range(n):
        path = sc.straight_path(p.path_steps, p.path_step_size)
        path[:, 5] += np.float32(0.4 * k) - 0.2 * np.sin(np.arange(p.path_steps) / (5.0 + k)).astype(np.float32)   # its own speed limits
        jobs.append((sc.initial_state(0.5 * k, 4.0 + 1.1 * k, 0.2 * (k % 3) - 0.2), path, k % len(envs)))
    got = []
    for k, (init, path, e) in enumerate(jobs):
        if k > 0:
            got.append(_traj_np(dp.collect_download((k - 1) & 1)[0]))
        dp.upload([init], path)
        dp.plan_resident(envs[e])
        dp.stage_download(k & 1)
    got.append(_traj_np(dp.collect_download((n - 1) & 1)[0]))
    assert dp.info()["fwd_overlap"] == 1, dp.info()
    for k, (init, path, e) in enumerate(jobs):
        tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occs[e], [init])
        assert np.array_equal(got[k].view(np.uint32), otrajs[0].view(np.uint32)), k
    assert len({g.tobytes() for g in got}) == n, "the jobs are not different plans"


def test_two_threads_with_their_own_engines():
    """Different engine objects may be driven from different threads (own non-blocking streams, thread-local error
    slot and bounce buffer; ctypes releases the GIL during the calls): two threads, each with its own environment
    and planner on a different grid, 12 plans each, concurrently -- every plan equals the oracle's."""
    import threading
    cases = [(201, 37, 20, 6, sc.OBSTACLES_OVERTAKE_TRAFFIC, 7.0), (128, 24, 16, 5, sc.OBSTACLES_OVERTAKE, 10.0)]
    want, jobs = [], []
    for S, V, A, T, obst, v0 in cases:
        p = sc.lon_params(S, V, A, T)
        ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
        path = sc.straight_path(p.path_steps, p.path_step_size)
        occ = sc.occ_map(ep, obst)
        inits = [sc.initial_state(1.0 * k, v0 + 0.25 * k, 0.0) for k in range(12)]
        _, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
        want.append(otrajs)
        jobs.append((p, ep, path, occ, inits))
    got, errors = [None, None], []
    start = threading.Barrier(2)

    def work(i):
        try:
            p, ep, path, occ, inits = jobs[i]
            env = tpl_amd.DynProgEnvironment()
            env.reinit_buffers(ep)
            env.set_occ_map(occ)
            dp = tpl_amd.DynProgLonPlanner()
            dp.reinit_buffers(p)
            start.wait()
            out = []
            for ini in inits:
                out.append(_traj_np(dp.update(ini, path, env)))
                env.get_occ_map()                            # a staged device-to-host copy between the plans
            got[i] = out
        except Exception as e:                               # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for i in range(2):
        assert got[i] is not None, f"thread {i} did not finish"
        for k, tr in enumerate(got[i]):
            assert np.array_equal(tr.view(np.uint32), want[i][k].view(np.uint32)), (i, k)


def test_unbounded_inputs_take_the_two_pass_kernel_and_still_match():
    """The two-stations-per-lane sweep keeps a running lexicographic argmin, which the host only
    selects for finite inputs below 1e6 (no NaN can arise); anything else runs the two-pass kernel.
    Both must reproduce the oracle bit for bit."""
    S, V, A, T = 256, 64, 40, 4            # large enough for the two-stations geometry
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    init = [sc.initial_state(0.0, 8.0, 0.0)]
    # (the third case is bounded but starts its arclength below zero: the two-station kernels read
    # the path's second tap as "the next slot", which needs s >= 0, so it takes the fallback too)
    for w_progress, s_min, bounded in ((0.1, 0.0, 1), (2.5e6, 0.0, 0), (0.1, -7.5, 0)):
        p = sc.lon_params(S, V, A, T)
        p.w_progress = w_progress
        p.s_min = s_min
        path = sc.straight_path(p.path_steps, p.path_step_size)
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        env.set_occ_map(occ)
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p)
        trajs = dp.update_batch(init, path, env)
        info = dp.info()
        assert info["big"] == 1 and info["bounded_inputs"] == bounded, info
        assert info["sweep"] == ("k_backward3v" if bounded else "k_backward"), info
        tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, init)
        _assert_tables_equal(dp, tables, T)
        assert np.array_equal(_traj_np(trajs[0]).view(np.uint32), otrajs[0].view(np.uint32))


# which sweep a plan under a non-default blend order (tplhip_dp_lon_set_blend_order) may take, by the kernel the default
# order takes on the same grid
BLEND_ORDER_KERNELS = {"k_backward3v": ("k_backward3o",), "k_backward3s": ("k_backward3o",), "k_backward": ("k_backward",)}


@pytest.mark.parametrize("name,quant,lerp,fma", [("wfloor", 2, 0, 1), ("wnone", 0, 0, 1), ("lerp1", 1, 1, 1), ("lerp2", 1, 2, 0),
                                                 ("wfloor_lerp1", 2, 1, 0), ("wfloor_lerp2", 2, 2, 1), ("wnone_lerp1", 0, 1, 0),
                                                 ("wnone_lerp2", 0, 2, 1), ("wfloor_unfused", 2, 0, 0)])
def test_filter_rule_variants_match_the_oracle_under_the_same_rule(name, quant, lerp, fma, monkeypatch):
    """R2 (weight quantisation) and R3 (blend form) stand in for the reference's texture unit; they are run-time
    parameters of the HIP planner (tplhip_dp_lon_set_filter_rule) and template parameters of the plain-fp32 sweeps,
    which are built for all nine combinations under both settings of R1.  Each non-default rule must reproduce the CPU
    restatement under the same rule (ONE checker library, dpo_set_rule) bit for bit -- tables, trajectories,
    re-evaluation -- on a small and on a several-rounds-deep grid, ON THE SAME KERNEL as the default rule, with the
    Newton form of the divisions, with IEEE divisions, and on the general sweep; setting the default again must bring
    back the default result."""
    for (S, V, A, T), want_default in (((201, 37, 20, 6), "k_backward3s"), ((128, 64, 66, 3), "k_backward3v")):
        p = sc.lon_params(S, V, A, T)
        ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
        path = sc.straight_path(p.path_steps, p.path_step_size)
        occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
        inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(4, seed=3)
        cp, cep = O.cparams(p), O.cparams(ep)
        with O.rule(quant=quant, lerp=lerp, fma=fma):
            tables, otrajs, _ = O.update(cp, cep, path, occ, inits)
        for dbg, want_kernel, want_forms in (("", want_default, (5, 5)), ("no_fastdiv", want_default, (0, 0)),
                                             ("no_v3,no_s3", "k_backward", None)):
            monkeypatch.setenv("TPLHIP_DEBUG", dbg)
            env = tpl_amd.DynProgEnvironment()
            env.reinit_buffers(ep)
            env.set_occ_map(occ)
            dp = tpl_amd.DynProgLonPlanner()
            dp.reinit_buffers(p)
            dp.set_contraction(bool(fma))
            dp.set_filter_rule(quant, lerp)
            assert dp.info()["rule"] == {"quant": quant, "lerp": lerp, "fma": fma, "order": 0}, dp.info()
            trajs = dp.update_batch(inits, path, env)
            info = dp.info()
            assert info["sweep"] == want_kernel, info
            if want_forms:
                assert (info["path_div3"], info["range_div3"]) == want_forms, info
            _assert_tables_equal(dp, tables, T)
            for r, tr in enumerate(trajs):
                assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (name, r)
        monkeypatch.setenv("TPLHIP_DEBUG", "")
        # the default rule again: default tables, default kernel
        dp.set_filter_rule(1, 0)
        dp.set_contraction(True)
        trajs = dp.update_batch(inits, path, env)
        assert dp.info()["sweep"] == "k_backward", dp.info()       # (this handle was created under no_v3,no_s3)
        tables0, otrajs0, _ = O.update(cp, cep, path, occ, inits)
        _assert_tables_equal(dp, tables0, T)
        assert any(not np.array_equal(tables0[t].view(np.uint32), tables[t].view(np.uint32)) for t in range(1, T)), "the variant changes nothing here"
    with pytest.raises(ValueError):
        dp.set_filter_rule(3, 0)


@pytest.mark.parametrize("name,quant,lerp", [("unfused", 1, 0), ("unfused_wfloor", 2, 0)])
def test_unfused_arithmetic_matches_the_oracle_under_r1_0(name, quant, lerp, monkeypatch):
    """R1 as a run-time choice (tplhip_dp_lon_set_contraction).  The default (since round 6) is the contraction nvcc applies
    to the reference's device code, written out as fma() on both sides -- no compiler decides what fuses; R1 = 0 is one
    rounding per operation of the source.  The HIP planner with contraction OFF reproduces the restatement under
    dpo_set_rule(., ., 0, .) bit for bit (distance map, tables, rollouts, re-evaluation) on a small and a
    several-rounds-deep grid -- on the SAME plain-fp32 sweeps as the default, with the same choice of division forms for the
    default filter rule (both R1 settings are built with every form), with IEEE divisions, and on the general sweep; switching
    it on again restores the default tables."""
    for (S, V, A, T), want_default, dt_start in (((201, 37, 20, 6), "k_backward3s", None), ((128, 64, 66, 3), "k_backward3v", 0.4)):
        p = sc.lon_params(S, V, A, T, dt_start=dt_start)
        ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31, dt_start=dt_start)
        path = sc.straight_path(p.path_steps, p.path_step_size)
        path[:, 2] *= 0.97                                              # stations lag the distance: interpPath blends for real
        path[:, 5] += 0.3 * np.sin(np.arange(p.path_steps) / 7.0).astype(np.float32)
        occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
        inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(4, seed=3)
        cp, cep = O.cparams(p), O.cparams(ep)
        with O.rule(quant=quant, lerp=lerp, fma=0):
            tables, otrajs, dist = O.update(cp, cep, path, occ, inits)
        ref = tpl_amd.DynProgLonPlanner()                               # the default rule's own choice of division forms on this grid
        ref.reinit_buffers(p)
        default_forms = (ref.info()["path_div3"], ref.info()["range_div3"])
        del ref
        for dbg, want_kernel, want_forms in (("", want_default, default_forms if (quant, lerp) == (1, 0) else (5, 5)),
                                             ("no_fastdiv", want_default, (0, 0)), ("no_v3,no_s3", "k_backward", None)):
            monkeypatch.setenv("TPLHIP_DEBUG", dbg)
            env = tpl_amd.DynProgEnvironment()
            env.reinit_buffers(ep)
            env.set_occ_map(occ)
            dp = tpl_amd.DynProgLonPlanner()
            dp.reinit_buffers(p)
            dp.set_contraction(False)
            dp.set_filter_rule(quant, lerp)
            trajs = dp.update_batch(inits, path, env)
            info = dp.info()
            assert info["sweep"] == want_kernel, info
            if want_forms:
                assert (info["path_div3"], info["range_div3"]) == want_forms, info
            assert np.array_equal(env.get_dist_map_path().view(np.uint32), dist.view(np.uint32))
            _assert_tables_equal(dp, tables, T)
            for r, tr in enumerate(trajs):
                assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (name, dbg, r)
            for r, tr in enumerate(trajs):                              # re-costing a plan gives the forward pass' numbers
                again = _traj_np(dp.reeval_traj(tr, path, env))
                assert np.array_equal(again[:, 5:].view(np.uint32), otrajs[r][:, 5:].view(np.uint32)), (name, dbg, "reeval", r)
        monkeypatch.setenv("TPLHIP_DEBUG", "")
        # on again (and the default filter rule): default tables
        dp.set_contraction(True)
        dp.set_filter_rule(1, 0)
        trajs = dp.update_batch(inits, path, env)
        tables0, otrajs0, _ = O.update(cp, cep, path, occ, inits)
        _assert_tables_equal(dp, tables0, T)
        assert any(not np.array_equal(tables0[t].view(np.uint32), tables[t].view(np.uint32)) for t in range(1, T)), "contraction changes nothing here"
    with pytest.raises(ValueError):
        dp.set_contraction(2)


@pytest.mark.parametrize("name,order,quant,lerp,fma", [("order_sva", 1, 1, 0, 1), ("sum8_f32", 2, 1, 0, 1), ("sum8_wide", 3, 1, 0, 1),
                                                        ("unfused_order_sva", 1, 1, 0, 0), ("r210o1", 1, 2, 1, 0),
                                                        ("r020o1", 1, 0, 2, 0), ("r001o2", 2, 0, 0, 1), ("r200o3", 3, 2, 0, 0)])
def test_blend_order_variants_match_the_oracle_under_the_same_rule(name, order, quant, lerp, fma, monkeypatch):
    """R3's ORDER and form as a run-time choice (tplhip_dp_lon_set_blend_order): sequential blends s -> v -> a instead of
    a -> v -> s, and the symmetric eight-corner sum (fp32 accumulator / wide accumulator) the texture hardware's
    documentation gives.  Each must reproduce the CPU restatement under the same rule (dpo_set_rule(., ., ., n), alone
    and combined with the other rules) bit for bit: distance map, tables, rollouts, re-evaluation, on a small and on a
    several-rounds-deep grid; setting order 0 again restores the default tables AND the default (plain-fp32) kernel."""
    for (S, V, A, T), want_default, dt_start in (((201, 37, 20, 6), "k_backward3s", None), ((128, 64, 66, 3), "k_backward3v", 0.4)):
        p = sc.lon_params(S, V, A, T, dt_start=dt_start)
        ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31, dt_start=dt_start)
        path = sc.straight_path(p.path_steps, p.path_step_size)
        path[:, 2] *= 0.97
        path[:, 5] += 0.3 * np.sin(np.arange(p.path_steps) / 7.0).astype(np.float32)
        occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
        inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(4, seed=3)
        cp, cep = O.cparams(p), O.cparams(ep)
        with O.rule(quant=quant, lerp=lerp, fma=fma, order=order):
            tables, otrajs, dist = O.update(cp, cep, path, occ, inits)
        for dbg in ("", "no_front", "no_v3,no_s3"):
            monkeypatch.setenv("TPLHIP_DEBUG", dbg)
            env = tpl_amd.DynProgEnvironment()
            env.reinit_buffers(ep)
            env.set_occ_map(occ)
            dp = tpl_amd.DynProgLonPlanner()
            dp.reinit_buffers(p)
            dp.set_contraction(bool(fma))
            dp.set_filter_rule(quant, lerp)
            dp.set_blend_order(order)
            trajs = dp.update_batch(inits, path, env)
            info = dp.info()
            # (k_backward3o is built for the weight rules rint and floor; un-quantised weights take the general sweep)
            assert info["sweep"] in BLEND_ORDER_KERNELS[want_default if "no_v3" not in dbg and quant != 0 else "k_backward"], info
            assert np.array_equal(env.get_dist_map_path().view(np.uint32), dist.view(np.uint32))
            _assert_tables_equal(dp, tables, T)
            for r, tr in enumerate(trajs):
                assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (name, dbg, r)
            for r, tr in enumerate(trajs):
                again = _traj_np(dp.reeval_traj(tr, path, env))
                assert np.array_equal(again[:, 5:].view(np.uint32), otrajs[r][:, 5:].view(np.uint32)), (name, dbg, "reeval", r)
        monkeypatch.setenv("TPLHIP_DEBUG", "")
        # order 0 on a fresh handle under the same other rules: the a -> v -> s tables of that build, on the plain-fp32 sweep
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p)
        dp.set_contraction(bool(fma))
        dp.set_filter_rule(quant, lerp)
        dp.set_blend_order(order)
        dp.set_blend_order(0)
        with O.rule(quant=quant, lerp=lerp, fma=fma):
            tables0, otrajs0, _ = O.update(cp, cep, path, occ, inits)
        trajs = dp.update_batch(inits, path, env)
        assert dp.info()["sweep"] == want_default, dp.info()
        _assert_tables_equal(dp, tables0, T)
        assert any(not np.array_equal(tables0[t].view(np.uint32), tables[t].view(np.uint32)) for t in range(1, T)), "the order changes nothing here"
    with pytest.raises(ValueError):
        dp.set_blend_order(4)


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("S,V,A,T,dt_start", [(128, 64, 64, 3, None), (96, 10, 64, 4, 0.4), (201, 37, 20, 4, None)])
def test_blend_order_with_trivial_a_blends(S, V, A, T, dt_start, order):
    """k_backward3o requests and parks only the a0 rows of an action whose blend along a is trivial (ACT_SINGLE_A: both a-rows
    clamped to the same row, or a weight of exactly 0) and reads half the taps.  64 acceleration rows put the jerk grid on the
    acceleration grid (weight exactly 0 for three of the nine actions) next to the clamped ones; 20 rows have the clamped kind
    only.  Full tables and rollouts against the oracle under the same order, raw bits."""
    p = sc.lon_params(S, V, A, T, dt_start=dt_start)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31, dt_start=dt_start)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.random_obstacles(6, 9))
    inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(3, seed=7)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    dp.set_blend_order(order)
    trajs = dp.update_batch(inits, path, env)
    assert dp.info()["sweep"] == "k_backward3o", dp.info()
    with O.rule(order=order):
        tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
    _assert_tables_equal(dp, tables, T)
    for r, tr in enumerate(trajs):
        assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), r


@pytest.mark.parametrize("dt_start", [None, 0.3, 0.6])
def test_forward_rollout_forms_agree(dt_start, monkeypatch):
    """The rollout runs with one step of speculation (eight wavefronts) where the table set fits the Infinity Cache and
    one step per round (one wavefront) beyond; TPLHIP_DEBUG=no_fwd_spec forces the second form on a small grid.  Both
    against the oracle, raw bits -- including dt_start values for which the terminal test of lon_planner.cu:93 fires
    in the middle of a speculated pair or not at all, odd and even stage counts."""
    for T in (5, 6):
        p = sc.lon_params(96, 12, 9, T, dt_start=dt_start)
        ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31, dt_start=dt_start)
        path = sc.straight_path(p.path_steps, p.path_step_size)
        occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
        inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(5, seed=21)
        for form in ("", "no_fwd_spec"):
            monkeypatch.setenv("TPLHIP_DEBUG", form)
            dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
            for r, tr in enumerate(trajs):
                assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (T, form, r)


@pytest.mark.parametrize("env_s,obst", [(201, sc.OBSTACLES_OVERTAKE), (64, ()), (130, sc.OBSTACLES_OVERTAKE_TRAFFIC),
                                        (1000, sc.random_obstacles(6, 11)), (9000, sc.OBSTACLES_OVERTAKE)])
def test_plan_front_forms_agree(env_s, obst, monkeypatch):
    """Distance map, stage prelude and terminal stage run as ONE launch (k_plan_front; every block builds the distance
    slices it needs from occupancy bits) or, with TPLHIP_DEBUG=no_front and where two slices do not fit the LDS of a
    block (9 000 environment stations), as the three kernels it stands for.  Both against the oracle, raw bits:
    distance map (no obstacle at all = the 10 000-chain over the whole row, bit rows of 1 to 141 words), tables,
    trajectories -- with dt_start values that put a stage between two time slices."""
    S, V, A, T = 96, 10, 8, 5
    for dt_start in (None, 0.3):
        p = sc.lon_params(S, V, A, T, dt_start=dt_start)
        ep = sc.env_params(t_steps=T, s_steps=env_s, l_steps=21, dt_start=dt_start)
        path = sc.straight_path(p.path_steps, p.path_step_size)
        occ = sc.occ_map(ep, obst)
        inits = [sc.initial_state(0.0, 8.0, 0.0), sc.initial_state(30.0, 3.0, 0.5)]
        for form in ("", "no_front"):
            monkeypatch.setenv("TPLHIP_DEBUG", form)
            dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
            assert dp.info()["plan_front"] == ("three launches" if form or env_s > 5000 else "k_plan_front")
            assert np.array_equal(env.get_dist_map_path().view(np.uint32), dist.view(np.uint32)), (env_s, form)
            _assert_tables_equal(dp, tables, T)
            for r, tr in enumerate(trajs):
                assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (env_s, form, r)


@pytest.mark.parametrize("dt_start", [None, 0.3, 0.6])
def test_one_stage_plan_forms_agree(dt_start, monkeypatch):
    """t_steps = 1: the reference's sweep loop (lon_planner.cu:347) does not run, so no table is ever written and the
    single rollout step reads table 0 as allocated (zeros) -- with a dt_start for which its terminal test does not fire
    the zeros are really read.  The fused front, whose terminal role would fill that table, must not be taken: both
    forms against the oracle, raw bits (the advisor's round-3 finding)."""
    p = sc.lon_params(96, 10, 8, 1, dt_start=dt_start)
    ep = sc.env_params(t_steps=2, s_steps=201, l_steps=21, dt_start=dt_start)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    inits = [sc.initial_state(0.0, 8.0, 0.0), sc.initial_state(30.0, 3.0, 0.5)]
    got = {}
    for form in ("", "no_front"):
        monkeypatch.setenv("TPLHIP_DEBUG", form)
        dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
        assert dp.info()["plan_front"] == "three launches", dp.info()
        for r, tr in enumerate(trajs):
            assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (form, r)
        got[form] = [_traj_np(tr) for tr in trajs]
    for a, b in zip(got[""], got["no_front"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_reeval_after_resident_plans_waits_for_their_rollouts():
    """Back-to-back mode: the rollout of the last plan_resident() runs on its own stream and reads the path; reeval_traj
    replaces that path.  It must wait for the pending rollouts first (the advisor's round-2 finding): two plans in
    flight, then a re-evaluation on ANOTHER path, then the staged result of the second plan -- all three against the
    oracle, raw bits."""
    S, V, A, T = 256, 64, 40, 6                     # large enough for the back-to-back (second table set) mode
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    other_path = path.copy()
    other_path[:, 5] = 9.5                          # another speed limit: the re-evaluation must not leak into the plan
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    init = sc.initial_state(0.0, 8.0, 0.0)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    dp.upload([init], path)
    dp.plan_resident(env)
    dp.plan_resident(env)
    dp.stage_download(0)
    cp, cep = O.cparams(p), O.cparams(ep)
    tables, otrajs, _ = O.update(cp, cep, path, occ, [init])
    # re-evaluate the oracle's plan on the other path while the rollouts may still be running
    traj = tpl_amd.LonTraj()
    traj.states = [tpl_amd.LonState() for _ in range(T)]
    for st, row in zip(traj.states, otrajs[0]):
        st.t, st.s, st.v, st.a, st.j, st.cost, st.constr = (float(x) for x in row)
    got_re = _traj_np(dp.reeval_traj(traj, other_path, env))
    got_plan = _traj_np(dp.collect_download(0)[0])
    assert np.array_equal(got_plan.view(np.uint32), otrajs[0].view(np.uint32))
    # the oracle re-costs the same states on the other path with the tables of the plan (lon_planner.cu:403-470)
    dist2 = O.dist_map_path(cep, occ, other_path, p.path_steps, p.path_step_size)
    out = (TplLonState * T)()
    rc = O.lib().dpo_reeval(C.byref(cp), C.byref(cep), O.fptr(other_path), O.fptr(dist2), O.fptr(tables), traj._as_array(), T, out)
    assert rc == 0
    assert np.array_equal(got_re.view(np.uint32), O.states_to_np(out).view(np.uint32))


# ------------------------------------------------------------------------------------------------------------------------
# Chained sweep (SweepChain in dp_lon_internal.h): the stages of a plan behind the terminal one in ONE launch
# ------------------------------------------------------------------------------------------------------------------------
CHAIN_CASES = [
    # (one residency round of workgroups per stage or more: 8 192 wavefronts on 256 compute units)
    pytest.param(256, 64, 64, 7, sc.OBSTACLES_OVERTAKE_TRAFFIC, None, id="rows-of-256-seven-stages"),
    pytest.param(128, 128, 64, 5, sc.random_obstacles(9, 7), 0.4, id="rows-of-128-dt_start"),
    pytest.param(128, 65, 128, 5, sc.random_obstacles(5, 6), None, id="odd-v-steps"),
    pytest.param(512, 64, 32, 5, sc.random_obstacles(7, 5), None, id="rows-of-512"),
]


@pytest.mark.parametrize("S,V,A,T,obst,dt_start", CHAIN_CASES)
def test_chained_sweep_matches_the_oracle_and_the_stage_by_stage_launches(S, V, A, T, obst, dt_start):
    """`update_batch` is one plan at a time: where the grid allows it (k_backward3v, rows of whole cache lines, four stages or
    more) its T - 2 non-terminal sweeps go out as ONE launch whose workgroups wait for the row pairs of the stage before
    them.  EVERY stage table and every trajectory against the oracle as raw bits; then the same plan stage by stage
    (`set_chained_sweep(0)`): no chained launch, the same bits."""
    p = sc.lon_params(S, V, A, T, dt_start=dt_start)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31, dt_start=dt_start)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, obst)
    inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(4, seed=11)
    dp, env, trajs, tables, otrajs, dist = _run_both(p, ep, path, occ, inits)
    info = dp.info()
    assert info["sweep"] == "k_backward3v" and info["chained_sweep"] == 1 and info["chained_launches"] == 1, info
    _assert_tables_equal(dp, tables, T)
    for r, tr in enumerate(trajs):
        assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), r
    dp.set_chained_sweep(0)
    again = dp.update_batch(inits, path, env)
    info = dp.info()
    assert info["chained_sweep"] == 0 and info["chained_launches"] == 1, info
    _assert_tables_equal(dp, tables, T)
    for r, tr in enumerate(again):
        assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), r


def test_chained_sweep_is_on_request_for_plans_in_flight():
    """`plan_resident` (plans back to back, possibly beside another planner's) sweeps stage by stage unless asked:
    `set_chained_sweep(1)` chains those too.  Four plans with a new start state and obstacle set each, back to back on one
    planner, against the oracle's trajectories; the launch counter tells which form ran."""
    S, V, A, T = 256, 64, 64, 5
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    cp, cep = O.cparams(p), O.cparams(ep)
    for mode, want_chained in ((-1, 0), (1, 4)):
        dp.set_chained_sweep(mode)
        before = dp.info()["chained_launches"]
        for k in range(4):
            occ = sc.occ_map(ep, sc.random_obstacles(3 + k, 5 + k))
            inits = sc.random_initial_states(2, seed=20 + k)
            env.set_occ_map(occ)
            dp.upload(inits, path)
            dp.plan_resident(env)
            got = dp.download()
            _, otrajs, _ = O.update(cp, cep, path, occ, inits)
            for r, tr in enumerate(got):
                assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (mode, k, r)
        info = dp.info()
        assert info["chained_launches"] - before == want_chained, (mode, info)
        assert info["chained_sweep"] == (2 if mode == 1 else 1), info


def test_a_chained_sweep_that_gives_up_fails_the_call_and_the_handle_recovers(monkeypatch):
    """No wait inside the chained launch is unbounded: a workgroup that gives up flags it, the call that hands out the plan
    fails loudly, and the handle sweeps stage by stage from then on.  TPLHIP_DEBUG=chain_fail_once makes the host read
    that flag as set once (the device side cannot be provoked: workgroups start in the order of their ids)."""
    monkeypatch.setenv("TPLHIP_DEBUG", "chain_fail_once")
    S, V, A, T = 256, 64, 64, 5
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    inits = [sc.initial_state(0.0, 8.0, 0.0)]
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    with pytest.raises(RuntimeError, match="plan again"):
        dp.update_batch(inits, path, env)
    assert dp.info()["chained_sweep"] == 0
    trajs = dp.update_batch(inits, path, env)
    tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
    assert dp.info()["chained_launches"] == 1
    _assert_tables_equal(dp, tables, T)
    assert np.array_equal(_traj_np(trajs[0]).view(np.uint32), otrajs[0].view(np.uint32))


def test_a_stage_that_leaves_half_the_chip_empty_is_not_chained():
    """Below one residency round of workgroups per stage the chained launch LOSES (the waiting workgroups of later stages take
    the free slots and every stage runs packed on part of the chip: 256x40x52 14.2 -> 19.1 us per stage sweep), and rows that
    are not whole cache lines cannot be chained at all: both sweep stage by stage, whatever the caller asks for."""
    for S, V, A, T in ((256, 40, 52, 5), (300, 128, 64, 5)):
        p = sc.lon_params(S, V, A, T)
        ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        env.set_occ_map(sc.occ_map(ep, sc.OBSTACLES_OVERTAKE))
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p)
        dp.set_chained_sweep(1)
        dp.update(sc.initial_state(0.0, 8.0, 0.0), sc.straight_path(p.path_steps, p.path_step_size), env)
        info = dp.info()
        assert info["sweep"] == "k_backward3v" and info["chained_sweep"] == 0 and info["chained_launches"] == 0, info


@pytest.mark.parametrize("debug,path_step", [("big=1,chain_any", 0.7), ("big=1,chain_any", 1.3), ("big=1,chain_any,no_fastdiv", 1.0), ("big=1,chain_any,skew=1", 2.0)])
def test_chained_sweep_under_every_division_form(debug, path_step, monkeypatch):
    """The chained form is a template instance per pair of division forms (reciprocal pair, Newton step, IEEE): a path step that is
    not a power of two / `no_fastdiv` select them, `big=1,chain_any` put a grid the oracle sweeps whole on the chained large-grid
    kernel; R1 = 0 as well (the other family of instances).  Every stage table against the oracle."""
    monkeypatch.setenv("TPLHIP_DEBUG", debug)
    S, V, A, T = 128, 21, 14, 6
    for fma in (1, 0):
        p = sc.lon_params(S, V, A, T)
        p.path_step_size = path_step
        p.path_steps = int(np.ceil(p.s_max / path_step)) + 5
        ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
        path = sc.straight_path(p.path_steps, p.path_step_size)
        occ = sc.occ_map(ep, sc.random_obstacles(12, 6))
        inits = [sc.initial_state(0.0, 8.0, 0.0)] + sc.random_initial_states(2, seed=3)
        env = tpl_amd.DynProgEnvironment()
        env.reinit_buffers(ep)
        env.set_occ_map(occ)
        dp = tpl_amd.DynProgLonPlanner()
        dp.reinit_buffers(p)
        dp.set_contraction(bool(fma))
        trajs = dp.update_batch(inits, path, env)
        with O.rule(fma=fma):
            tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
        info = dp.info()
        assert info["sweep"] == "k_backward3v" and info["chained_launches"] == 1, info
        _assert_tables_equal(dp, tables, T)
        for r, tr in enumerate(trajs):
            assert np.array_equal(_traj_np(tr).view(np.uint32), otrajs[r].view(np.uint32)), (fma, r)


def test_a_chained_launch_whose_flags_never_come_drains_and_is_reported(monkeypatch):
    """The device side of the same safety net: under TPLHIP_DEBUG=chain_stall the workgroups of the next chained launch wait for
    flag values nobody writes.  Every one of them gives up (after 20 us of the 100 MHz counter, or at once when it sees that
    another has), the launch drains, the status word in page-locked memory says so, the call fails, and the next plan -- swept
    stage by stage -- matches the oracle."""
    monkeypatch.setenv("TPLHIP_DEBUG", "chain_stall")
    S, V, A, T = 256, 64, 64, 5
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE)
    inits = [sc.initial_state(0.0, 8.0, 0.0)]
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    with pytest.raises(RuntimeError, match="plan again"):
        dp.update_batch(inits, path, env)
    info = dp.info()
    assert info["chained_launches"] == 1 and info["chained_failures"] == 1 and info["chained_sweep"] == 0, info
    trajs = dp.update_batch(inits, path, env)
    tables, otrajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, inits)
    assert dp.info()["chained_launches"] == 1
    _assert_tables_equal(dp, tables, T)
    assert np.array_equal(_traj_np(trajs[0]).view(np.uint32), otrajs[0].view(np.uint32))
