"""Occupancy pipeline: tpl_amd.DynProgEnvironment on the GPU against
oracle/dp_env_oracle.c, bit for bit, through the C-ABI of libtplhip.so."""
import numpy as np
import pytest

import oracle_env as oe
from tpl_amd import DynProgEnvironment, DynProgLonPlanner
from benchkit import scenarios as sc

pytestmark = pytest.mark.gpu


def _run_both(ep, rl, step, objs):
    env = DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_ref_line(rl, step)
    o = oe.OracleEnv(ep)
    o.set_ref_line(rl, step)
    for g, st in objs:
        env.insert_geometry(g, st)
        o.insert_geometry(g, st)
    env.update()
    ref = o.update()
    return env, ref


def _assert_same(env, ref):
    img, occ, dl = ref
    got_img = env.get_occ_map_cartesian()
    assert got_img.shape == img.shape
    assert np.array_equal(got_img, img), f"{np.count_nonzero(got_img != img)} pixels differ"
    assert np.array_equal(env.get_occ_map(), occ)
    assert np.array_equal(env.get_dist_map_lon(), dl)
    # the published form of the two tables (dp_env.py:185-188), transposed on the device
    assert np.array_equal(env.get_occ_map_slices(), np.vstack(np.moveaxis(occ, 2, 0)))
    assert np.array_equal(env.get_dist_map_lon_slices(), np.vstack(np.moveaxis(dl, 2, 0)))


@pytest.mark.parametrize("kind,seed", [("straight", 0), ("straight", 3), ("arc", 1), ("arc", 7)])
def test_scene_parity(kind, seed):
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = sc.env_scene(kind, seed=seed, n_objects=8)
    env, ref = _run_both(ep, rl, step, objs)
    assert np.count_nonzero(ref[0]) > 0
    _assert_same(env, ref)
    assert env.ref_line.x_offset == pytest.approx((rl[:, 0].min() + rl[:, 0].max()) / 2)


def test_other_grid_and_dilation():
    ep = sc.env_params(t_steps=31, s_steps=128, l_steps=21)
    ep.s_max, ep.l_min, ep.l_max, ep.dilation = 150.0, -4.0, 6.0, 0.6
    rl, step, objs = sc.env_scene("arc", seed=11, n_objects=12, t_steps=31, dt=0.25)
    ep.dt, ep.dt_start = 0.25, 0.25
    env, ref = _run_both(ep, rl, step, objs)
    _assert_same(env, ref)


def test_random_triangles_offscreen_degenerate_partial():
    """Rasteriser edge cases: slivers, zero-area and huge triangles, vertices far outside the image,
    a trailing partial triangle."""
    ep = sc.env_params(t_steps=8, s_steps=201, l_steps=31)
    rl = sc.ref_line_straight(x0=0.0, y0=0.0, heading=0.0)
    rng = np.random.default_rng(5)
    tri = np.zeros((3 * 400 + 2, 3))
    tri[:, 0] = rng.uniform(-30.0, 260.0, len(tri))
    tri[:, 1] = rng.uniform(-140.0, 140.0, len(tri))
    tri[:, 2] = np.repeat(rng.integers(0, 8, len(tri) // 3 + 1), 3)[:len(tri)].astype(float)
    tri[30:60, 1] = 0.25 * np.round(tri[30:60, 1] * 4)           # vertices on pixel corners
    tri[30:60, 0] = 0.25 * np.round(tri[30:60, 0] * 4) + 109.75
    tri[60:63, :2] = tri[60, :2]                                  # a point
    tri[63:66, 1] = 7.0                                           # a line
    tri[66:69, :2] = [(-5000.0, -9000.0), (9000.0, -9000.0), (100.0, 9000.0)]   # covers the whole image
    tri[69:72, :2] = [(1e7, 1e7), (1e7 + 1, 1e7), (1e7, 1e7 + 1)]               # far away
    env, ref = _run_both(ep, rl, 0.5, [(tri[:600], False), (tri[600:], True)])
    _assert_same(env, ref)


def test_empty_scene_and_repeat():
    ep = sc.env_params(t_steps=5, s_steps=64, l_steps=9)
    ep.s_max = 100.0
    rl, step, objs = sc.env_scene("straight", seed=2)
    env, ref = _run_both(ep, rl, step, [])
    _assert_same(env, ref)
    assert np.count_nonzero(env.get_occ_map_cartesian()) == 0
    # geometry is consumed by update(): a second update sees an empty scene again
    env.insert_geometry(objs[0][0], False)
    env.update()
    env.update()
    _assert_same(env, ref)


def test_errors_like_the_reference():
    env = DynProgEnvironment()
    assert env.get_occ_map_cartesian().shape == (1, 1)
    with pytest.raises(RuntimeError, match="cannot insert dyn object points with ref_line of 0 steps"):
        env.insert_geometry(np.zeros((3, 3)), False)
    with pytest.raises(ValueError, match="incorrect shape"):
        env.insert_geometry(np.zeros((3, 2)), False)
    with pytest.raises(ValueError, match="incorrect shape"):
        env.set_ref_line(np.zeros((10, 8)), 0.5)
    env.set_ref_line(sc.ref_line_straight(n=100), 0.5)          # 50 m < default s_max 200 m
    with pytest.raises(RuntimeError, match="refline length = 50.000000 < than environment s_max = 200.000000"):
        env.update()
    assert env.get_dist_map_dir(0).shape == (env.params.dir_steps, env.params.l_steps, env.params.s_steps)


def test_filtered_occupancy_lookup():
    ep = sc.env_params(t_steps=6, s_steps=101, l_steps=21)
    ep.s_max = 100.0
    rl, step, objs = sc.env_scene("straight", seed=4)
    env, ref = _run_both(ep, rl, step, objs)
    occ = ref[1]
    # on cell centres the filtered lookup returns the cell (interpDistField, env.cu:225-234)
    for (il, is_, it) in [(10, 50, 0), (3, 20, 2), (20, 100, 5), (0, 0, 0)]:
        t = 0.0 if it == 0 else ep.dt_start + (it - 1) * ep.dt
        s = ep.s_min + is_ * (ep.s_max - ep.s_min) / (ep.s_steps - 1)
        l = ep.l_min + il * (ep.l_max - ep.l_min) / (ep.l_steps - 1)
        assert env.get_dist(t, s, l) == pytest.approx(float(occ[il, is_, it]), abs=1e-3)


def test_planner_on_rendered_scene_matches_oracle_plan():
    """End to end: geometry -> GPU occupancy pipeline -> GPU DP plan == oracle pipeline -> oracle plan."""
    import oracle_dp
    p = sc.lon_params(s_steps=64, v_steps=24, a_steps=12, t_steps=10)
    ep = sc.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl = sc.ref_line_straight(x0=0.0, y0=0.0, heading=0.0)
    ts = np.array([0.0] + [1.0 + i for i in range(9)])
    objs = [(sc.prediction_soup(rl, 0.5, 60.0, 2.0, 4.5, -1.0, 1.0, ts), False),
            (sc.prediction_soup(rl, 0.5, 140.0, 0.0, 4.5, -1.0, 1.0, ts), True)]
    env, ref = _run_both(ep, rl, 0.5, objs)
    _assert_same(env, ref)
    path = sc.straight_path()
    init = sc.initial_state(v=10.0)
    dp = DynProgLonPlanner()
    dp.reinit_buffers(p)
    traj = dp.update(init, path, env)
    _, otrajs, _ = oracle_dp.update(oracle_dp.cparams(p), oracle_dp.cparams(ep), path, ref[1], [init])
    want = otrajs[0]
    got = np.array([[getattr(s, f) for f in ("t", "s", "v", "a", "j", "cost", "constr")] for s in traj.states],
                   dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got[:, 6].max() == 0.0 and got[-1, 1] > 30.0          # a feasible plan that moves


def test_unusual_parameters_and_bit_31():
    """s_min > 0, asymmetric lateral range, wide dilation, 33 time slices: slice 31 shares its bit with
    the stationary flag and slices >= 32 have no bit at all (rule G3) -- oracle and GPU must agree on
    what that means."""
    ep = sc.env_params(t_steps=33, s_steps=150, l_steps=17)
    ep.s_min, ep.s_max, ep.l_min, ep.l_max, ep.dilation = 10.0, 160.0, -2.5, 6.0, 2.4
    ep.dt, ep.dt_start = 0.2, 0.1
    rl = sc.ref_line_arc(n=400, step=0.5, radius=80.0, semantic_range=(60, 200))
    ts = np.array([0.0] + [0.1 + 0.2 * i for i in range(32)])
    objs = [(sc.prediction_soup(rl, 0.5, 40.0, 6.0, 4.5, -1.0, 1.0, ts), False),
            (sc.prediction_soup(rl, 0.5, 90.0, 0.0, 4.5, 1.0, 3.0, ts[:3]), True),
            (sc.prediction_soup(rl, 0.5, 120.0, 1.0, 4.5, -2.0, 0.0, ts), False)]
    env, ref = _run_both(ep, rl, 0.5, objs)
    _assert_same(env, ref)
    occ = ref[1]
    assert occ.shape == (17, 150, 33)
    assert (occ[:, :, 32] == 1.0).sum() == 0          # no bit for slice 32
    assert (occ == 3.0).any() and (occ[:, :, 5] == 1.0).any()


@pytest.mark.gpu
def test_long_station_axis_narrows_the_distance_map_tiles():
    """450 stations: the LDS tiles of the distance-map scans hold 11 time slices instead of 16, and
    25 slices make three tiles with a ragged last one."""
    ep = sc.env_params(t_steps=25, s_steps=450, l_steps=13)
    rl, step, objs = sc.env_scene("arc", seed=5, n_objects=6)
    env, ref = _run_both(ep, rl, step, objs)
    _assert_same(env, ref)
