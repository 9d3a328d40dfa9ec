#!/usr/bin/env python3
"""Generates tests/golden/dp_planner.npz by running the REFERENCE's planner plugin classes
(tpl.planning.dyn_prog.poly_lat_dp_lon_planner.PolyLatDpLonPlanner with its DpEnv, imported from /root/reference, which
exists in the build container only) over the synthetic drives of tests/dp_planner_cases.py.

    python tests/golden/make_dp_planner_goldens.py

What runs is the reference's own Python -- parameter handling, reference-line and prediction preparation, the shift of
the previous plan, the replan decision, path assembly, trajectory assembly, the emergency counter.  What it CALLS is
shimmed, because tplcpp needs CUDA and cannot be built here:
  * tplcpp     -> tests/oracle_tplcpp.py: tplcpp's surface over this repository's CPU restatements of the engines
                  (oracle/*.c) and its host routines for project / resample / lerp / gen_prediction_geometry;
  * numba      -> ``njit`` is the identity decorator;   * structstore -> never called;
  * objtoolbox -> ``bundle()`` a namespace whose attributes come back as proxies with ``deepcopy()`` (what a structstore
                  struct gives the planner), ``merge(dst, src)`` copies src's attributes onto dst.
So the fixture pins the plugin LOGIC against the reference's Python, with the same engines on both sides; the
engines' own parity is the subject of the other tests."""
import copy
import os
import sys
import types
from contextlib import contextmanager
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)


class _Inert:
    def __getattr__(self, k):
        return _Inert()

    def __call__(self, *a, **k):
        return _Inert()

    def __gt__(self, o):
        return False

    def __lt__(self, o):
        return False

    def __sub__(self, o):
        return 0

    def __index__(self):
        return 0


class _Proxy:
    """What reading an object back from a structstore gives: attribute access plus deepcopy()."""

    def __init__(self, obj):
        object.__setattr__(self, "_obj", obj)

    def __getattr__(self, k):
        return getattr(self._obj, k)

    def __setattr__(self, k, v):
        setattr(self._obj, k, v)

    def deepcopy(self):
        return copy.deepcopy(self._obj)


class _Store(SimpleNamespace):
    def __setattr__(self, k, v):
        if hasattr(v, "__dict__") and not isinstance(v, (SimpleNamespace, _Proxy)):
            v = _Proxy(v)
        object.__setattr__(self, k, v)


def install_shims():
    import oracle_tplcpp
    from tpl_amd.planning import geometry as geo
    numba = types.ModuleType("numba")
    numba.njit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
    sys.modules["numba"] = numba
    otb = types.ModuleType("objtoolbox")
    otb.merge = lambda dst, src: [setattr(dst, k, getattr(src, k)) for k in vars(src) if not k.startswith("_")] and None
    otb.bundle = lambda: _Store()
    sys.modules["objtoolbox"] = otb
    sys.modules["structstore"] = types.ModuleType("structstore")
    tplcpp = oracle_tplcpp.module()
    tplcpp.lerp = geo.lerp_cpp
    fallback = _Inert()
    tplcpp.__getattr__ = lambda name: fallback
    sys.modules["tplcpp"] = tplcpp


@contextmanager
def _lock():
    yield


def set_dead_time(pl, seconds):
    """What a user does in tpl's GUI: the planner plans from `seconds` ahead, the environment shifts its slices."""
    pl.shared.params.planner.dead_time = seconds
    pl.shared.params.env.dead_time = seconds


def run_drive(make_planner, drive, cycles, closed_loop, dead_time=0.0):
    import dp_planner_cases as cases
    pl = make_planner()
    if dead_time:
        set_dead_time(pl, dead_time)
    rows, meta = [], []
    follow = None
    for k in range(cycles):
        traj = pl.update(drive.cycle(k, follow))
        r = cases.trajectory_row(traj)
        rows.append(np.zeros((0, 8)) if r is None else r)
        meta.append([float(traj.emergency), pl.last_replan_time, pl.dp_lon_start.s, pl.dp_lon_start.v, pl.dp_lon_start.a,
                     pl.poly_lat_start.s, pl.poly_lat_start.l, pl.dp_env.ref_line_shift, pl.dp_env.dt_start, pl.dt_start])
        if closed_loop and r is not None:
            # the vehicle sits where the plan said it would be one cycle later
            i = int(np.searchsorted(r[:, 0], r[0, 0] + cases.DT))
            follow = (r[i, 2] - np.cos(r[i, 6]) * 1.58, r[i, 3] - np.sin(r[i, 6]) * 1.58, r[i, 6], r[i, 4], r[i, 5])
    return rows, np.array(meta), pl


def run_drive_ll(make_planner, drive, cycles, dead_time=0.0):
    import dp_planner_cases as cases
    pl = make_planner()
    if dead_time:
        set_dead_time(pl, dead_time)
    rows, meta = [], []
    for k in range(cycles):
        traj = pl.update(drive.cycle(k, man_time=(k >= 4)))
        r = cases.trajectory_row(traj)
        rows.append(np.zeros((0, 8)) if r is None else r)
        s0 = pl.dp_planner.traj_smooth[0]
        meta.append([float(traj.emergency), pl.last_replan_time, s0.s, s0.ds, s0.l, pl.dp_env.ref_line_shift,
                     pl.dt_start, float(pl.emergency_counter), len(pl.traj_dp.states)])
    return rows, np.array(meta), pl


def main():
    install_shims()
    sys.path.insert(0, "/root/reference/library")
    import dp_planner_cases as cases
    from tpl.planning.dyn_prog.poly_lat_dp_lon_planner import PolyLatDpLonPlanner
    out = {}
    for name, kw, cycles, closed, dead in (("nominal", dict(v_ego=9.0), 14, False, 0.0), ("closed_loop", dict(v_ego=7.0), 14, True, 0.0),
                                           ("manual", dict(v_ego=11.0, automated=False), 12, False, 0.0),
                                           ("dead_time", dict(v_ego=8.0), 12, False, 0.25),
                                           ("crossing_reset_jump", dict(v_ego=12.0, crossing=True, reset_at=5, jump_at=9), 13, False, 0.0),
                                           ("blocked", dict(v_ego=9.0, blocked=True), 16, False, 0.0)):
        rows, meta, pl = run_drive(lambda: PolyLatDpLonPlanner(_Store(), _lock), cases.Drive(**kw), cycles, closed, dead)
        for k, r in enumerate(rows):
            out[f"{name}.traj_{k}"] = r
        out[f"{name}.meta"] = meta
        out[f"{name}.ref_line"] = pl.dp_env.ref_line
        out[f"{name}.path"] = pl.path
        out[f"{name}.occ"] = pl.dp_env.cpp_env.get_occ_map()
        print(name, [len(r) for r in rows], "replans at", sorted(set(np.round(meta[:, 1], 2))))
    from tpl.planning.dyn_prog.dp_lat_lon_planner import DpLatLonPlanner
    for name, kw, cycles, dead in (("ll_nominal", dict(v_ego=9.0), 12, 0.0), ("ll_manual", dict(v_ego=6.0, automated=False), 14, 0.0),
                                   ("ll_dead_time", dict(v_ego=8.0), 12, 0.25),
                                   ("ll_crossing_reset_jump", dict(v_ego=12.0, crossing=True, reset_at=5, jump_at=9), 13, 0.0),
                                   ("ll_blocked", dict(v_ego=9.0, blocked=True), 6, 0.0)):
        rows, meta, pl = run_drive_ll(lambda: DpLatLonPlanner(_Store(), _lock), cases.Drive(**kw), cycles, dead)
        for k, r in enumerate(rows):
            out[f"{name}.traj_{k}"] = r
        out[f"{name}.meta"] = meta
        out[f"{name}.traj_dp"] = pl.dp_planner.traj_dp.as_numpy()
        print(name, [len(r) for r in rows], "replans at", sorted(set(np.round(meta[:, 1], 2))))
    path = os.path.join(HERE, "dp_planner.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
