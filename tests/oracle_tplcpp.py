"""A module object with tplcpp's Python surface whose ENGINES are the CPU restatements under oracle/ (test
infrastructure: only tests/ and the fixture generators import it; the product never does).

It lets code written against tplcpp -- the reference's own planner classes in the build container, this repository's
mirrors of them in the CPU test-suite, tools/dump_cycle_goldens.py -- run without a GPU:

  DynProgEnvironment   -> oracle/dp_env_oracle.c         PolyLatPlanner       -> oracle/poly_lat_oracle.c
  DynProgLonPlanner    -> oracle/dp_lon_oracle.c         DynProgLatLonPlanner -> oracle/lat_lon_oracle.c

Parameter objects, states and trajectories are tpl_amd's own classes (ctypes structs + host routines of
libtplhip.so that never touch the GPU): they are containers, the same on both sides of every comparison."""
import ctypes as C
import types

import numpy as np

import oracle_dp as O
import oracle_env as oe
import oracle_lat_lon as oll
import oracle_poly_lat as opl
import tpl_amd
from tpl_amd._capi import TplLatLonCartState, TplLatLonState, TplLonState, TplPolyLatTrajPoint


class DynProgEnvironment:
    def __init__(self):
        self.params = tpl_amd.DynProgEnvParams()
        self.ref_line = tpl_amd.RefLine()
        self._o = None
        self._ref = None
        self._pending = []
        self._img = np.zeros((1, 1), np.uint32)
        self._occ = self._dl = None
        self.reinit_buffers(self.params, True)

    def reinit_buffers(self, params, force=False):
        self.params = params.__deepcopy__({})
        self._o = oe.OracleEnv(self.params)
        sh = (self.params.l_steps, self.params.s_steps, self.params.t_steps)
        if self._occ is None or self._occ.shape != sh or force:
            self._occ = np.zeros(sh, np.float32)
            self._dl = np.zeros(sh + (2,), np.float32)
        if self._ref is not None:
            self._o.set_ref_line(*self._ref)

    def set_ref_line(self, ref_line, step_size):
        ref_line = np.ascontiguousarray(ref_line, dtype=np.float64)
        if ref_line.ndim != 2 or ref_line.shape[1] != 9:
            raise ValueError(f"array has incorrect shape, expected: (-1, 9), got: {ref_line.shape}")
        self._ref = (ref_line.copy(), step_size)
        self._o.set_ref_line(ref_line, step_size)
        self.ref_line = tpl_amd.RefLine(self._o.ref.copy(), np.float32(step_size), float(self._o.offsets[0]),
                                        float(self._o.offsets[1]))

    def insert_geometry(self, geom, stationary):
        geom = np.ascontiguousarray(geom, dtype=np.float64)
        if geom.ndim != 2 or geom.shape[1] != 3:
            raise ValueError(f"array has incorrect shape, expected: (-1, 3), got: {geom.shape}")
        if self._ref is None:
            raise RuntimeError("cannot insert dyn object points with ref_line of 0 steps")
        self._pending.append((geom, bool(stationary)))

    def update(self):
        for g, st in self._pending:
            self._o.insert_geometry(g, st)
        self._pending = []
        res = self._o.update()
        if res is None:
            raise RuntimeError("refline length < than environment s_max")
        self._img, self._occ, self._dl = res

    def set_occ_map(self, occ):
        self._occ = np.ascontiguousarray(occ, dtype=np.float32)

    def get_occ_map_cartesian(self):
        return self._img

    def get_occ_map(self):
        return self._occ

    def get_dist_map_lon(self):
        return self._dl

    def get_dist_map_dir(self, idx_t):
        p = self.params
        return np.zeros((p.dir_steps, p.l_steps, p.s_steps), np.float32)


class PolyLatPlanner:
    def __init__(self):
        self.params = tpl_amd.PolyLatPlannerParams()
        self.best = (0, 0)

    def reinit_buffers(self, params):
        self.params = params.__deepcopy__({})

    def update(self, start, env):
        o = env._o
        _, best, pts, q = opl.plan(self.params, o.c, start, o.ref, o.step, o.offsets, env._occ)
        arr = (TplPolyLatTrajPoint * len(pts))()
        for i, row in enumerate(pts):
            for f, v in zip(opl.FIELDS, row):
                setattr(arr[i], f, v)
        traj = tpl_amd.PolyLatTraj()
        traj._take(arr)
        C.memmove(C.byref(traj.poly._c), C.byref(q), C.sizeof(q))
        self.best = best
        return traj


class DynProgLonPlanner:
    def __init__(self):
        self.params = tpl_amd.DynProgLonPlannerParams()
        self._tables = self._dist = None

    def reinit_buffers(self, params):
        self.params = params.__deepcopy__({})

    def _path(self, path):
        path = np.ascontiguousarray(path, dtype=np.float32)
        want = (self.params.path_steps, 7)
        if path.shape != want:
            raise ValueError(f"array has incorrect shape, expected: {want}, got: {path.shape}")
        return path

    def update(self, initial_state, path, env):
        p, ep = O.cparams(self.params), O.cparams(env.params)
        self._tables, trajs, self._dist = O.update(p, ep, self._path(path), env._occ, [initial_state])
        out = (TplLonState * p.t_steps)()
        for i, row in enumerate(trajs[0]):
            for (name, _), v in zip(TplLonState._fields_, row):
                setattr(out[i], name, float(v))
        return tpl_amd.LonTraj._from_array(out)

    def reeval_traj(self, traj, path, env):
        p, ep = O.cparams(self.params), O.cparams(env.params)
        if len(traj.states) != p.t_steps:
            raise RuntimeError("trajectory size does not match t_steps")
        path = self._path(path)
        dist = O.dist_map_path(ep, env._occ, path, p.path_steps, p.path_step_size)
        tables = self._tables if self._tables is not None else np.zeros(
            (p.t_steps, p.a_steps, p.v_steps, p.s_steps, 4), np.float32)
        out = (TplLonState * p.t_steps)()
        assert O.lib().dpo_reeval(C.byref(p), C.byref(ep), O.fptr(path), O.fptr(dist), O.fptr(tables), traj._as_array(),
                                  p.t_steps, out) == 0
        return tpl_amd.LonTraj._from_array(out)


class DynProgLatLonPlanner:
    def __init__(self):
        self.params = tpl_amd.DynProgLatLonPlannerParams()
        self.traj_dp = tpl_amd.LatLonTraj()
        self.traj_smooth = tpl_amd.LatLonTraj()
        self.traj_dp_cart = tpl_amd.LatLonCartTraj()
        self.traj_smooth_cart = tpl_amd.LatLonCartTraj()
        self.lat_dists = []
        self._tables = None
        self.reinit_buffers(self.params, True)

    def reinit_buffers(self, params, force=False):
        full = force or any(getattr(params, f) != getattr(self.params, f) for f in ("t_steps", "s_steps", "ds_steps", "l_steps"))
        self.params = params.__deepcopy__({})
        if full:                                            # lat_lon_planner.cu:539-541
            T = self.params.t_steps
            self.traj_dp.states = (self.traj_dp.states + [tpl_amd.LatLonState() for _ in range(T)])[:T]
            self.traj_smooth.states = self.traj_smooth.states[:1]
            self.traj_smooth_cart.states = self.traj_smooth_cart.states[:1]

    def update_traj_dp(self, env):
        o = env._o
        self._tables, otraj = oll.update_traj_dp(self.params, o.c, env._dl, o.ref, o.step, self.traj_dp[0])
        self.traj_dp._take(otraj)

    def update_traj_smooth(self, env):
        dp = self.traj_dp._as_array()
        cap = max(2, int(dp[len(dp) - 1].t / self.params.dt_smooth_traj) + 2)
        out = (TplLatLonState * cap)()
        n = oll.lib().llo_update_traj_smooth(C.byref(self.params._c), dp, len(dp), C.byref(self.traj_smooth[0]._c), out, cap)
        self.lat_dists = [tpl_amd.LatDistances() for _ in range(len(self.traj_smooth.states))]
        self.traj_smooth._take(out, n)

    def frenet_to_cartesian(self, traj, ref_line):
        arr = traj._as_array()
        cart = (TplLatLonCartState * len(arr))()
        pts = np.ascontiguousarray(ref_line.as_array(), dtype=np.float64)
        off = np.array([ref_line.x_offset, ref_line.y_offset], dtype=np.float64)
        dp = C.POINTER(C.c_double)
        oll.lib().llo_frenet_to_cartesian(pts.ctypes.data_as(dp), len(pts), np.float32(ref_line.step_size),
                                          off.ctypes.data_as(dp), arr, len(arr), cart)
        res = tpl_amd.LatLonCartTraj()
        res._take(cart)
        return res

    def update_traj_cart(self, ref_line):
        self.traj_dp_cart = self.frenet_to_cartesian(self.traj_dp, ref_line)
        self.traj_smooth_cart = self.frenet_to_cartesian(self.traj_smooth, ref_line)

    def reeval_traj(self, traj, env):
        o = env._o
        out = oll.reeval(self.params, o.c, env._dl, o.ref, o.step, traj._as_array())
        res = tpl_amd.LatLonTraj()
        res._take(out)
        return res

    def copy_backward_pass_tex(self, idx_t):
        return self._tables[idx_t]


def module():
    """The module object: engines above, everything else tpl_amd's."""
    m = types.ModuleType("oracle_tplcpp")
    for name in ("DynProgEnvParams", "PolyLatPlannerParams", "PolyLatTrajPoint", "PolyLatTraj", "PolyQuintic",
                 "DynProgLonPlannerParams", "LonState", "LonTraj", "lon_dynamics", "DynProgLatLonPlannerParams",
                 "LatLonState", "LatLonCartState", "LatLonTraj", "LatLonCartTraj", "LatDistances", "RefLine",
                 "RefLinePoint", "Projection", "project", "resample", "gen_prediction_geometry"):
        setattr(m, name, getattr(tpl_amd, name))
    m.DynProgEnvironment = DynProgEnvironment
    m.PolyLatPlanner = PolyLatPlanner
    m.DynProgLonPlanner = DynProgLonPlanner
    m.DynProgLatLonPlanner = DynProgLatLonPlanner
    m.cuda_get_device_count = lambda: 0
    m.cuda_set_device = lambda idx: None
    return m
