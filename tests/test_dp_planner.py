"""The planner plugins around the DP engines -- tpl_amd.planning.dyn_prog.{DpEnv, PolyLatDpLonPlanner} -- against
the REFERENCE's own classes.  tests/golden/dp_planner.npz holds, cycle by cycle, what
tpl.planning.dyn_prog.poly_lat_dp_lon_planner.PolyLatDpLonPlanner returned on the drives of tests/dp_planner_cases.py
when its engine calls went to tests/oracle_tplcpp.py (generator: tests/golden/make_dp_planner_goldens.py).

CPU: the mirror on the same engines must reproduce every cycle exactly (plugin logic pinned).
GPU: the mirror on the HIP engines must reproduce them within the engines' own bars (DP bit-exact, lattice 1e-5)."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

import dp_planner_cases as cases
from tpl_amd.planning.dyn_prog import DpLatLonPlanner, PolyLatDpLonPlanner
from tpl_amd.store import Bundle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "dp_planner.npz"))
DRIVES = {"nominal": (dict(v_ego=9.0), 14, False, 0.0), "closed_loop": (dict(v_ego=7.0), 14, True, 0.0),
          "manual": (dict(v_ego=11.0, automated=False), 12, False, 0.0),
          "dead_time": (dict(v_ego=8.0), 12, False, 0.25),      # plans start 0.25 s ahead, bridged by the previous trajectory
          # an intersecting road with traffic (stop line flags, crossing-traffic sweep by ego speed and object
          # acceleration), a reset request at cycle 5, the vehicle found 3 m off its plan at cycle 9
          "crossing_reset_jump": (dict(v_ego=12.0, crossing=True, reset_at=5, jump_at=9), 13, False, 0.0),
          # no valid plan for more than ten cycles: the planner declares an emergency (no trajectory) and restarts
          "blocked": (dict(v_ego=9.0, blocked=True), 16, False, 0.0)}


@contextmanager
def _lock():
    yield


def _run(name, engines, rtol, atol):
    kw, cycles, closed, dead_time = DRIVES[name]
    pl = PolyLatDpLonPlanner(Bundle(), _lock, engines=engines)
    if dead_time:
        pl.shared.params.planner.dead_time = pl.shared.params.env.dead_time = dead_time
    drive = cases.Drive(**kw)
    follow = None
    want_meta = GOLD[f"{name}.meta"]
    for k in range(cycles):
        traj = pl.update(drive.cycle(k, follow))
        r = cases.trajectory_row(traj)
        want = GOLD[f"{name}.traj_{k}"]
        assert (r is None) == (len(want) == 0), f"cycle {k}: emergency flag differs"
        if r is not None:
            assert r.shape == want.shape, f"cycle {k}: {r.shape} vs {want.shape}"
            assert np.allclose(r, want, rtol=rtol, atol=atol), \
                f"{name} cycle {k}: trajectory differs by {np.max(np.abs(r - want)):.3e}"
        meta = np.array([float(traj.emergency), pl.last_replan_time, pl.dp_lon_start.s, pl.dp_lon_start.v,
                         pl.dp_lon_start.a, pl.poly_lat_start.s, pl.poly_lat_start.l, pl.dp_env.ref_line_shift,
                         pl.dp_env.dt_start, pl.dt_start])
        assert np.allclose(meta, want_meta[k], rtol=rtol, atol=atol), f"{name} cycle {k}: planner state {meta} vs {want_meta[k]}"
        if closed and r is not None:
            # follow the GOLDEN plan so that both sides see the same vehicle states even at the looser GPU bar
            i = int(np.searchsorted(want[:, 0], want[0, 0] + cases.DT))
            follow = (want[i, 2] - np.cos(want[i, 6]) * 1.58, want[i, 3] - np.sin(want[i, 6]) * 1.58, want[i, 6],
                      want[i, 4], want[i, 5])
    assert np.allclose(pl.dp_env.ref_line, GOLD[f"{name}.ref_line"], rtol=rtol, atol=atol)
    assert np.allclose(pl.path, GOLD[f"{name}.path"], rtol=max(rtol, 1e-12), atol=max(atol, 1e-12))
    return pl


@pytest.mark.parametrize("name", sorted(DRIVES))
def test_plugin_logic_matches_the_reference_classes(name):
    import oracle_tplcpp
    pl = _run(name, oracle_tplcpp.module(), rtol=0.0, atol=1e-12)
    assert np.array_equal(pl.dp_env.cpp_env.get_occ_map(), GOLD[f"{name}.occ"])
    # the debug block the reference publishes every cycle
    dbg = pl.shared.debug
    assert dbg.planner.traj_lon.shape[1] == 4 and dbg.planner.traj_lat.shape[1] == 12 and dbg.planner.path.shape[1] == 7
    p = pl.dp_env.cpp_env.params
    assert dbg.env.occ_map.shape == (p.t_steps * p.l_steps, p.s_steps) and dbg.env.ref_line.shape[1] == 9


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DRIVES))
def test_hip_engines_under_the_plugin(name):
    import tpl_amd
    pl = _run(name, None, rtol=1e-5, atol=1e-5)
    assert isinstance(pl.dp_lon, tpl_amd.DynProgLonPlanner) and isinstance(pl.dp_env.cpp_env, tpl_amd.DynProgEnvironment)
    assert np.array_equal(pl.dp_env.cpp_env.get_occ_map(), GOLD[f"{name}.occ"])


LL_DRIVES = {"ll_nominal": (dict(v_ego=9.0), 12, 0.0), "ll_manual": (dict(v_ego=6.0, automated=False), 14, 0.0),
             "ll_dead_time": (dict(v_ego=8.0), 12, 0.25),
             "ll_crossing_reset_jump": (dict(v_ego=12.0, crossing=True, reset_at=5, jump_at=9), 13, 0.0),
             "ll_blocked": (dict(v_ego=9.0, blocked=True), 6, 0.0)}


def _run_ll(name, engines, rtol, atol):
    kw, cycles, dead_time = LL_DRIVES[name]
    pl = DpLatLonPlanner(Bundle(), _lock, engines=engines)
    if dead_time:
        pl.shared.params.planner.dead_time = pl.shared.params.env.dead_time = dead_time
    drive = cases.Drive(**kw)
    want_meta = GOLD[f"{name}.meta"]
    for k in range(cycles):
        traj = pl.update(drive.cycle(k, man_time=(k >= 4)))
        r = cases.trajectory_row(traj)
        want = GOLD[f"{name}.traj_{k}"]
        assert (r is None) == (len(want) == 0), f"cycle {k}: emergency flag differs"
        if r is not None:
            assert r.shape == want.shape, f"{name} cycle {k}: {r.shape} vs {want.shape}"
            assert np.allclose(r, want, rtol=rtol, atol=atol), f"{name} cycle {k}: differs by {np.max(np.abs(r - want)):.3e}"
        s0 = pl.dp_planner.traj_smooth[0]
        meta = np.array([float(traj.emergency), pl.last_replan_time, s0.s, s0.ds, s0.l, pl.dp_env.ref_line_shift, pl.dt_start,
                         float(pl.emergency_counter), len(pl.traj_dp.states)])
        assert np.allclose(meta, want_meta[k], rtol=rtol, atol=atol), f"{name} cycle {k}: planner state {meta} vs {want_meta[k]}"
    assert np.allclose(pl.dp_planner.traj_dp.as_numpy(), GOLD[f"{name}.traj_dp"], rtol=rtol, atol=atol)
    return pl


@pytest.mark.parametrize("name", sorted(LL_DRIVES))
def test_lat_lon_plugin_logic_matches_the_reference_class(name):
    import oracle_tplcpp
    pl = _run_ll(name, oracle_tplcpp.module(), rtol=0.0, atol=1e-9)
    dbg = pl.shared.debug.planner
    assert dbg.traj_dp.shape[1] == 9 and dbg.traj_smooth_cart.shape[1] == 8
    # the drive with the localisation jump went through the "too far from the plan" restart and says so
    assert dbg.reinit_msg.startswith("Warning: Planner reinit") == ("jump" in name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(LL_DRIVES))
def test_hip_lat_lon_engine_under_the_plugin(name):
    import tpl_amd
    pl = _run_ll(name, None, rtol=1e-6, atol=1e-6)
    assert isinstance(pl.dp_planner, tpl_amd.DynProgLatLonPlanner)


def test_skips_when_time_stands_still_and_handles_missing_map():
    import oracle_tplcpp
    pl = PolyLatDpLonPlanner(Bundle(), _lock, engines=oracle_tplcpp.module())
    drive = cases.Drive()
    sh = drive.cycle(0)
    t0 = pl.update(sh)
    replan = pl.last_replan_time
    t1 = pl.update(sh)                                 # same env.t: no update (poly_lat_dp_lon_planner.py:432-435)
    assert t1 is t0 and pl.last_replan_time == replan
    sh2 = drive.cycle(1)
    sh2.local_map = None                               # :429-430
    assert pl.update(sh2) is t0
