"""Python surface of the DP (s,v,a) longitudinal planner -- the part of the
reference's pybind11 module ``tplcpp`` that HP-1 defines
(library/src/python/bindings_dp_lon.cu:7-137, bindings_dp_env.cu:7-106).

Same class names, attribute names, argument meaning and error behaviour, so
``tpl.planning.dyn_prog.PolyLatDpLonPlanner`` can call these in place of
``tplcpp.DynProgLonPlanner`` & friends.  All compute runs
in ``libtplhip.so`` on the GPU; nothing here computes on the CPU.
"""

import ctypes as C
import sys
import weakref

import numpy as np

from . import _capi
from ._capi import TplDpEnvParams, TplDpLonParams, TplLonState, TplRefLinePoint, check


def _fptr(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_float))


class _StructView:
    """Attribute access to a ctypes struct + the ``__slots__`` enumeration the
    reference's param classes expose for otb.merge / structstore
    (bindings_dp_lon.cu:36-64)."""

    _ctype = None
    _slot_names = ()

    def __init__(self):
        object.__setattr__(self, "_c", self._ctype())
        self._set_defaults()

    def _set_defaults(self):
        pass

    def __getattr__(self, name):
        c = object.__getattribute__(self, "_c")
        if name in type(self)._all_fields():
            return getattr(c, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in type(self)._all_fields():
            setattr(self._c, name, value)
        else:
            raise AttributeError(f"{type(self).__name__} has no attribute {name!r}")

    @classmethod
    def _all_fields(cls):
        names = cls.__dict__.get("_field_names")
        if names is None:
            names = frozenset(f[0] for f in cls._ctype._fields_)
            cls._field_names = names
        return names

    def __deepcopy__(self, memo):
        other = type(self)()
        C.memmove(C.byref(other._c), C.byref(self._c), C.sizeof(self._c))
        return other

    def __copy__(self):
        return self.__deepcopy__({})

    def __eq__(self, other):
        return type(other) is type(self) and bytes(self._c) == bytes(other._c)

    def __repr__(self):
        body = ", ".join(f"{n}={getattr(self._c, n)!r}" for n in self._slot_names)
        return f"{type(self).__name__}({body})"


# pybind exposes ``__slots__`` as a read-only *instance* property; a property named
# ``__slots__`` cannot be written in the class body (type() would consume it), so it
# is attached afterwards.
_StructView.__slots__ = property(lambda self: list(self._slot_names))


class DynProgLonPlannerParams(_StructView):
    """tplcpp.DynProgLonPlannerParams (bindings_dp_lon.cu:9-64)."""
    _ctype = TplDpLonParams
    _slot_names = (
        "s_min", "s_max", "v_min", "v_max", "a_min", "a_max", "j_min", "j_max",
        "t_steps", "s_steps", "v_steps", "a_steps", "dt", "dt_start",
        "time_gap", "gap_min", "w_progress", "w_a", "w_j", "w_snap", "w_safety_dist",
        "path_step_size", "path_steps", "width_veh", "length_veh")

    def _set_defaults(self):
        _capi.load().tplhip_dp_lon_params_default(C.byref(self._c))


class DynProgEnvParams(_StructView):
    """tplcpp.DynProgEnvParams (bindings_dp_env.cu:9-51)."""
    _ctype = TplDpEnvParams
    _slot_names = (
        "t_min", "t_max", "l_min", "l_max", "s_min", "s_max", "ds_max", "dir_min", "dir_max",
        "scale_objects", "dilation", "dt", "dt_start", "t_steps", "l_steps", "s_steps", "dir_steps")

    def _set_defaults(self):
        _capi.load().tplhip_dp_env_params_default(C.byref(self._c))


class LonState(_StructView):
    """tplcpp.LonState (bindings_dp_lon.cu:66-83)."""
    _ctype = TplLonState
    _slot_names = ("t", "s", "v", "a", "j", "cost", "constr")

    @classmethod
    def _from_c(cls, c):
        st = cls()
        C.memmove(C.byref(st._c), C.byref(c), C.sizeof(TplLonState))
        return st

    def as_numpy(self):
        c = self._c
        return np.array([c.s, c.v, c.a, c.j], dtype=np.float64)


def lon_dynamics(x0, j, dt):
    """tplcpp.lon_dynamics (bindings_dp_lon.cu:85; lon_planner.cu:240-250)."""
    out = TplLonState()
    _capi.load().tplhip_lon_dynamics(C.byref(x0._c), float(j), float(dt), C.byref(out))
    return LonState._from_c(out)


class LonTraj:
    """tplcpp.LonTraj (bindings_dp_lon.cu:87-118)."""

    def __init__(self):
        self.states = []

    @classmethod
    def _from_array(cls, arr):
        traj = cls()
        traj.states = [LonState._from_c(c) for c in arr]
        return traj

    def _as_array(self):
        arr = (TplLonState * len(self.states))()
        for i, s in enumerate(self.states):
            C.memmove(C.byref(arr[i]), C.byref(s._c), C.sizeof(TplLonState))
        return arr

    def copy(self):
        return LonTraj._from_array(self._as_array())

    def state(self, t):
        lib = _capi.load()
        arr = self._as_array()
        if np.ndim(t) == 0:
            out = TplLonState()
            check(lib.tplhip_lon_traj_state(arr, len(self.states), float(t), C.byref(out)))
            return LonState._from_c(out)
        ts = np.asarray(t, dtype=np.float64)
        if ts.ndim != 1:
            raise ValueError(f"array has incorrect shape, expected: (-1,), got: {ts.shape}")
        ts = np.ascontiguousarray(ts)
        res = np.empty((ts.shape[0], 4), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        check(lib.tplhip_lon_traj_state_rows(arr, len(self.states), ts.ctypes.data_as(dp), len(ts), res.ctypes.data_as(dp)))
        return res

    def as_numpy(self):
        res = np.empty((len(self.states), 4), dtype=np.float64)
        for i, s in enumerate(self.states):
            res[i] = (s._c.s, s._c.v, s._c.a, s._c.j)
        return res


class RefLinePoint(_StructView):
    """tplcpp.RefLinePoint (bindings_dp_common.cu:9-19)."""
    _ctype = TplRefLinePoint
    _slot_names = tuple(f[0] for f in TplRefLinePoint._fields_)


class RefLine:
    """tplcpp.RefLine (bindings_dp_common.cu:21-36; utils.hpp:176-220): the centred points, the step size and the
    offsets that were subtracted.  ``points`` reads as a list of RefLinePoint like the reference's vector; the
    float64[n, 9] array behind it is ``as_array()``."""

    def __init__(self, points=None, step_size=0.0, x_offset=0.0, y_offset=0.0, owner=None):
        self._array = np.zeros((0, 9)) if points is None else np.ascontiguousarray(points, dtype=np.float64)
        self.step_size = step_size
        self.x_offset = x_offset
        self.y_offset = y_offset
        # the environment that holds this line on the device (weak: the line must not keep it alive)
        self._owner = weakref.ref(owner) if owner is not None else None

    @staticmethod
    def fromarray(arr, step_size, no_offset=False):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if arr.ndim != 2 or arr.shape[1] != 9:
            raise ValueError(f"array has incorrect shape, expected: (-1, 9), got: {arr.shape}")
        out = np.empty_like(arr)
        xo, yo = C.c_double(), C.c_double()
        dp = C.POINTER(C.c_double)
        check(_capi.load().tplhip_ref_line_center(arr.ctypes.data_as(dp), len(arr), 1 if no_offset else 0,
                                                  out.ctypes.data_as(dp), C.byref(xo), C.byref(yo)))
        return RefLine(out, np.float32(step_size), xo.value, yo.value)

    def as_array(self):
        return self._array

    @property
    def points(self):
        pts = []
        for row in self._array:
            p = RefLinePoint()
            C.memmove(C.byref(p._c), row.ctypes.data, C.sizeof(TplRefLinePoint))
            pts.append(p)
        return pts

    @points.setter
    def points(self, value):
        if isinstance(value, np.ndarray):
            self._array = np.ascontiguousarray(value, dtype=np.float64).reshape(-1, 9)
        else:
            self._array = np.array([[getattr(p, f) for f in RefLinePoint._slot_names] for p in value],
                                   dtype=np.float64).reshape(-1, 9)
        self._owner = None                      # no longer the line its environment holds

    def lerp(self, s):
        """RefLine::lerp (utils.hpp:217-219) -> RefLinePoint at station s."""
        out = RefLinePoint()
        dp = C.POINTER(C.c_double)
        check(_capi.load().tplhip_ref_line_lerp(self._array.ctypes.data_as(dp), len(self._array), float(self.step_size),
                                                float(s), C.cast(C.byref(out._c), dp)))
        return out


class _ResultPool:
    """Result arrays whose memory is recycled once every holder has dropped them.  The environment plugin fetches a
    2.8 MB image and two tables of 0.25 / 0.5 MB every cycle; a fresh numpy array of that size is fresh pages, ~700
    page faults while it is filled.  An array is handed out again only when nothing but the pool refers to it (views
    keep their base alive and count), so a caller cannot tell it from a new one."""

    def __init__(self, depth=8):                 # three kinds of result per cycle, each alternating between two buffers
        self._depth = depth
        self._arrays = []
        # What sys.getrefcount reports for an array that ONLY the pool holds, seen from inside take(): an implementation
        # detail of the interpreter (list + local + argument = 3 on CPython 3.10; newer interpreters count fewer), so it
        # is measured on a throw-away array through the same code path instead of assumed.  No match -> no recycling.
        self._idle = None
        probe = np.empty(1, dtype=np.uint8)
        self._arrays.append(probe)
        del probe
        self._idle = self._refs(0)
        held = self._arrays[0]                   # one outside holder must read exactly one more, or the count is useless
        if self._refs(0) != self._idle + 1:
            self._idle = None
        del held
        self._arrays.clear()

    def _refs(self, k):
        a = self._arrays[k]
        return sys.getrefcount(a)

    def take(self, shape, dtype):
        if self._idle is not None:
            for k in range(len(self._arrays)):
                a = self._arrays[k]
                if a.shape == shape and a.dtype == dtype and sys.getrefcount(a) == self._idle:
                    return a
                del a
        a = np.empty(shape, dtype=dtype)
        self._arrays.append(a)
        del self._arrays[:-self._depth]
        return a


class DynProgEnvironment:
    """tplcpp.DynProgEnvironment (bindings_dp_env.cu:53-105).

    ``set_ref_line`` + ``insert_geometry`` + ``update`` run the occupancy pipeline
    (env.cu:352-437) on the GPU: the triangle soup is rasterised by a compute
    kernel instead of OpenGL (tpl_amd/csrc/dp_env.hip).  ``set_occ_map`` is an
    addition: it hands a Frenet occupancy map over directly, in the layout
    ``get_occ_map`` returns (``[l_steps, s_steps, t_steps]`` float32).
    """

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        self._lib = _capi.load()
        check(self._lib.tplhip_dp_env_create(int(device), C.byref(self._h)))
        self._results = _ResultPool()
        self.params = DynProgEnvParams()
        self.ref_line = RefLine()
        self.reinit_buffers(self.params, True)          # env.cu:282-285

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.tplhip_dp_env_destroy(h)
            self._h = None

    def reinit_buffers(self, params, force=False):
        check(self._lib.tplhip_dp_env_reinit(self._h, C.byref(params._c)))
        self.params = params.__deepcopy__({})

    def _shape(self):
        p = self.params
        return (p.l_steps, p.s_steps, p.t_steps)

    def set_occ_map(self, occ):
        occ = np.ascontiguousarray(occ, dtype=np.float32)
        if occ.shape != self._shape():
            raise ValueError(f"array has incorrect shape, expected: {self._shape()}, got: {occ.shape}")
        check(self._lib.tplhip_dp_env_set_occ_map(self._h, _fptr(occ)))

    def update_dist_maps(self):
        """updateDistMapLon (env.cu:95-129) alone, on the occupancy set by set_occ_map / the last update()."""
        check(self._lib.tplhip_dp_env_update_dist_maps(self._h))

    def get_occ_map(self):
        occ = np.empty(self._shape(), dtype=np.float32)
        check(self._lib.tplhip_dp_env_get_occ_map(self._h, _fptr(occ)))
        return occ

    def get_dist_map_path(self):
        p = self.params
        dist = np.empty((p.t_steps, p.s_steps), dtype=np.float32)
        check(self._lib.tplhip_dp_env_get_dist_map_path(self._h, _fptr(dist)))
        return dist

    # ---- occupancy pipeline, bindings_dp_env.cu:64-105 ----

    def set_ref_line(self, ref_line, step_size):
        ref_line = np.ascontiguousarray(ref_line, dtype=np.float64)
        if ref_line.ndim != 2 or ref_line.shape[1] != 9:
            raise ValueError(f"array has incorrect shape, expected: (-1, 9), got: {ref_line.shape}")
        check(self._lib.tplhip_dp_env_set_ref_line(self._h, ref_line.ctypes.data_as(C.POINTER(C.c_double)),
                                                   ref_line.shape[0], float(step_size)))
        xo, yo = C.c_double(), C.c_double()
        check(self._lib.tplhip_dp_env_ref_line_offset(self._h, C.byref(xo), C.byref(yo)))
        pts = ref_line.copy()
        pts[:, 0] -= xo.value
        pts[:, 1] -= yo.value
        self.ref_line = RefLine(pts, np.float32(step_size), xo.value, yo.value, owner=self)

    def insert_geometry(self, geom, stationary):
        geom = np.ascontiguousarray(geom, dtype=np.float64)
        if geom.ndim != 2 or geom.shape[1] != 3:
            raise ValueError(f"array has incorrect shape, expected: (-1, 3), got: {geom.shape}")
        check(self._lib.tplhip_dp_env_insert_geometry(self._h, geom.ctypes.data_as(C.POINTER(C.c_double)),
                                                      geom.shape[0], 1 if stationary else 0))

    def update(self):
        check(self._lib.tplhip_dp_env_update(self._h))

    def get_occ_map_cartesian(self):
        n = C.c_int()
        check(self._lib.tplhip_dp_env_cart_size(self._h, C.byref(n)))
        img = self._results.take((n.value, n.value), np.dtype(np.uint32))
        check(self._lib.tplhip_dp_env_get_occ_map_cartesian(self._h, img.ctypes.data_as(C.POINTER(C.c_uint32))))
        return img

    def get_dist_map_lon(self):
        out = np.empty(self._shape() + (2,), dtype=np.float32)
        check(self._lib.tplhip_dp_env_get_dist_map_lon(self._h, _fptr(out)))
        return out

    def get_occ_map_slices(self):
        """np.vstack(np.moveaxis(get_occ_map(), 2, 0)) -- the time slices below each other, [t_steps * l_steps, s_steps]
        -- transposed on the device (what dp_env.py:185-186 publishes every cycle)."""
        p = self.params
        out = self._results.take((p.t_steps * p.l_steps, p.s_steps), np.dtype(np.float32))
        check(self._lib.tplhip_dp_env_get_occ_map_slices(self._h, _fptr(out)))
        return out

    def get_dist_map_lon_slices(self):
        """np.vstack(np.moveaxis(get_dist_map_lon(), 2, 0)): [t_steps * l_steps, s_steps, 2] (dp_env.py:187-188)."""
        p = self.params
        out = self._results.take((p.t_steps * p.l_steps, p.s_steps, 2), np.dtype(np.float32))
        check(self._lib.tplhip_dp_env_get_dist_map_lon_slices(self._h, _fptr(out)))
        return out

    def get_dist_map_dir(self, idx_t):
        # the reference never launches updateDirDistMap (env.cu:160-217 has no caller), so the
        # array it hands out is the zero-initialised allocation
        p = self.params
        return np.zeros((p.dir_steps, p.l_steps, p.s_steps), dtype=np.float32)

    def get_dist(self, t, s, l, dt=0.0):
        out = C.c_float()
        check(self._lib.tplhip_dp_env_get_dist(self._h, float(t), float(s), float(l), C.byref(out)))
        return out.value


class DynProgLonPlanner:
    """tplcpp.DynProgLonPlanner (bindings_dp_lon.cu:120-136)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        self._lib = _capi.load()
        check(self._lib.tplhip_dp_lon_create(int(device), C.byref(self._h)))
        self.params = DynProgLonPlannerParams()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.tplhip_dp_lon_destroy(h)
            self._h = None

    def reinit_buffers(self, params):
        check(self._lib.tplhip_dp_lon_reinit(self._h, C.byref(params._c)))
        self.params = params.__deepcopy__({})

    def _path(self, path):
        # pybind force-casts the caller's float64 array to float32
        # (bindings_dp_lon.cu:125-128, binding_helpers.hpp:19-20)
        path = np.ascontiguousarray(path, dtype=np.float32)
        want = (self.params.path_steps, 7)
        if path.shape != want:
            raise ValueError(f"array has incorrect shape, expected: {want}, got: {path.shape}")
        return path

    def update(self, initial_state, path, env):
        path = self._path(path)
        T = self.params.t_steps
        out = (TplLonState * T)()
        check(self._lib.tplhip_dp_lon_update(self._h, C.byref(initial_state._c), _fptr(path), env._h, out))
        return LonTraj._from_array(out)

    def update_batch(self, initial_states, path, env):
        """n rollouts from n initial states over one backward sweep (new)."""
        path = self._path(path)
        n = len(initial_states)
        T = self.params.t_steps
        inits = (TplLonState * n)()
        for i, s in enumerate(initial_states):
            C.memmove(C.byref(inits[i]), C.byref(s._c), C.sizeof(TplLonState))
        out = (TplLonState * (n * T))()
        check(self._lib.tplhip_dp_lon_update_batch(self._h, n, inits, _fptr(path), env._h, out))
        return [LonTraj._from_array(out[i * T:(i + 1) * T]) for i in range(n)]

    def reeval_traj(self, traj, path, env):
        path = self._path(path)
        arr = traj._as_array()
        out = (TplLonState * max(1, len(traj.states)))()
        check(self._lib.tplhip_dp_lon_reeval(self._h, arr, len(traj.states), _fptr(path), env._h, out))
        return LonTraj._from_array(out)

    # ---- parity / measurement helpers (not in the reference) ------------------

    def copy_table(self, t):
        p = self.params
        out = np.empty((p.a_steps, p.v_steps, p.s_steps, 4), dtype=np.float32)
        check(self._lib.tplhip_dp_lon_copy_table(self._h, int(t), _fptr(out)))
        return out

    def upload(self, initial_states, path):
        path = self._path(path)
        n = len(initial_states)
        inits = (TplLonState * n)()
        for i, s in enumerate(initial_states):
            C.memmove(C.byref(inits[i]), C.byref(s._c), C.sizeof(TplLonState))
        check(self._lib.tplhip_dp_lon_upload(self._h, n, inits, _fptr(path)))
        self._n_up = n

    def plan_resident(self, env):
        check(self._lib.tplhip_dp_lon_plan_resident(self._h, env._h))

    def sync(self):
        check(self._lib.tplhip_dp_lon_sync(self._h))

    def download(self):
        n, T = self._n_up, self.params.t_steps
        out = (TplLonState * (n * T))()
        check(self._lib.tplhip_dp_lon_download(self._h, out))
        return [LonTraj._from_array(out[i * T:(i + 1) * T]) for i in range(n)]

    def stage_download(self, slot):
        check(self._lib.tplhip_dp_lon_stage_download(self._h, int(slot)))

    def collect_download(self, slot):
        n, T = self._n_up, self.params.t_steps
        out = (TplLonState * (n * T))()
        check(self._lib.tplhip_dp_lon_collect_download(self._h, int(slot), out))
        return [LonTraj._from_array(out[i * T:(i + 1) * T]) for i in range(n)]

    # ---- scenario batches (new) -----------------------------------

    def upload_scenarios(self, occ_maps, env_params, initial_states, paths, chunk=32, n=None):
        """n independent scenarios, resident on the device: ``occ_maps`` float32[n, l, s, t] (per scenario the
        layout of ``DynProgEnvironment.get_occ_map``), ``initial_states`` n LonStates, ``paths`` one
        float32[path_steps, 7] shared by all or float32[n, path_steps, 7].  ``chunk`` scenarios are swept per
        launch (value tables are allocated for that many).  ``occ_maps=None, initial_states=None, n=N`` reserves the
        device side for N scenarios; the maps then arrive block by block (``set_scenario_block``)."""
        ep = env_params
        reserve = occ_maps is None
        if reserve:
            if initial_states is not None or n is None:
                raise ValueError("reserving needs occ_maps=None, initial_states=None and n")
        else:
            occ = np.ascontiguousarray(occ_maps, dtype=np.float32)
            n = len(initial_states)
            want = (n, ep.l_steps, ep.s_steps, ep.t_steps)
            if occ.shape != want:
                raise ValueError(f"array has incorrect shape, expected: {want}, got: {occ.shape}")
        paths = np.ascontiguousarray(paths, dtype=np.float32)
        if paths.ndim == 2:
            paths = paths[None]
        if paths.shape[1:] != (self.params.path_steps, 7) or paths.shape[0] not in (1, n):
            raise ValueError(f"array has incorrect shape, expected: (1 or {n}, {self.params.path_steps}, 7), got: {paths.shape}")
        inits = None
        if not reserve:
            inits = (TplLonState * n)()
            for i, s in enumerate(initial_states):
                C.memmove(C.byref(inits[i]), C.byref(s._c), C.sizeof(TplLonState))
        check(self._lib.tplhip_dp_lon_scenarios_upload(self._h, n, None if reserve else _fptr(occ), C.byref(ep._c), inits,
                                                       _fptr(paths), paths.shape[0], int(chunk)))
        self._sc_n = n
        self._sc_ep = ep

    def set_scenario_block(self, first, occ_maps, initial_states):
        """Scenarios [first, first + len(initial_states)) of a reserved set: float32[count, l, s, t] and their states."""
        ep = self._sc_ep
        occ = np.ascontiguousarray(occ_maps, dtype=np.float32)
        count = len(initial_states)
        want = (count, ep.l_steps, ep.s_steps, ep.t_steps)
        if occ.shape != want:
            raise ValueError(f"array has incorrect shape, expected: {want}, got: {occ.shape}")
        inits = (TplLonState * count)()
        for i, s in enumerate(initial_states):
            C.memmove(C.byref(inits[i]), C.byref(s._c), C.sizeof(TplLonState))
        check(self._lib.tplhip_dp_lon_scenarios_set_block(self._h, int(first), count, _fptr(occ), inits))

    def plan_scenarios(self, first=0, count=None, id_base=0, reset_best=True):
        """Enqueue the plans of scenarios [first, first + count) (one launch per stage for all of them) and fold
        their best plan into the device-side record."""
        count = self._sc_n - first if count is None else count
        check(self._lib.tplhip_dp_lon_scenarios_plan(self._h, int(first), int(count), int(id_base), 1 if reset_best else 0))

    def download_scenarios(self):
        n, T = self._sc_n, self.params.t_steps
        out = (TplLonState * (n * T))()
        check(self._lib.tplhip_dp_lon_scenarios_download(self._h, out))
        return [LonTraj._from_array(out[i * T:(i + 1) * T]) for i in range(n)]

    def best_record_bytes(self):
        """The reduced record as it sits in device memory: TplDpPlanRecord + TplLonState[t_steps] (uint8 array)."""
        buf = np.zeros(self._lib.tplhip_dp_lon_record_bytes(self.params.t_steps), dtype=np.uint8)
        check(self._lib.tplhip_dp_lon_scenarios_best(self._h, buf.ctypes.data_as(C.c_void_p)))
        return buf

    def best_record(self):
        """The reduced record as numpy: (constr, cost, id, float32[t_steps, 7] states)."""
        buf = self.best_record_bytes()
        return (float(buf[0:4].view(np.float32)[0]), float(buf[4:8].view(np.float32)[0]), int(buf[8:12].view(np.int32)[0]),
                buf[16:].view(np.float32).reshape(self.params.t_steps, 7).copy())

    def best_record_device(self):
        """(device address, bytes) of the record -- for a collective that sends it straight from HBM."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        check(self._lib.tplhip_dp_lon_scenarios_best_device(self._h, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def copy_scenario_table(self, scenario, t):
        p = self.params
        out = np.empty((p.a_steps, p.v_steps, p.s_steps, 4), dtype=np.float32)
        check(self._lib.tplhip_dp_lon_scenarios_copy_table(self._h, int(scenario), int(t), _fptr(out)))
        return out

    def info(self):
        import json
        buf = C.create_string_buffer(512)
        check(self._lib.tplhip_dp_lon_info(self._h, buf, 512))
        return json.loads(buf.value.decode())

    def set_filter_rule(self, weight_quant=1, lerp_form=0):
        """R2 / R3 of the HP-1 arithmetic contract (the stand-ins for the reference's texture unit) as a run-time
        choice; ``(1, 0)`` is the default.  See ``tplhip_dp_lon_set_filter_rule`` in include/tplhip.h."""
        check(self._lib.tplhip_dp_lon_set_filter_rule(self._h, int(weight_quant), int(lerp_form)))

    def set_contraction(self, on=True):
        """R1 of the HP-1 arithmetic contract: the multiply-adds nvcc fuses by default (-fmad=true), written out as
        fma() calls.  ON is the default (it is how the reference's device code is built); ``False`` = one rounding per
        source operation.  See ``tplhip_dp_lon_set_contraction`` in include/tplhip.h."""
        check(self._lib.tplhip_dp_lon_set_contraction(self._h, int(on)))

    def set_blend_order(self, order=0):
        """R3's order / form: 0 = sequential blends a -> v -> s (default), 1 = s -> v -> a, 2 / 3 = the symmetric
        eight-corner sum with an fp32 / a wide accumulator.  See ``tplhip_dp_lon_set_blend_order`` in include/tplhip.h."""
        check(self._lib.tplhip_dp_lon_set_blend_order(self._h, int(order)))

    def set_chained_sweep(self, mode=-1):
        """All stages of a plan behind the terminal one in ONE launch of the large-grid sweep (no kernel boundary per
        stage): -1 = ``update`` / ``update_batch`` only (default), 1 = every plan incl. ``plan_resident``, 0 = never.
        Same bits; see ``tplhip_dp_lon_set_chained_sweep`` in include/tplhip.h for when it applies and what it costs a
        caller that keeps several planners busy."""
        check(self._lib.tplhip_dp_lon_set_chained_sweep(self._h, int(mode)))

    def timing_enable(self, enable=True):
        check(self._lib.tplhip_dp_lon_timing_enable(self._h, 1 if enable else 0))

    def timing_get(self):
        ms = C.c_double()
        n = C.c_int64()
        check(self._lib.tplhip_dp_lon_timing_get(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def cuda_get_device_count():
    """tplcpp.cuda_get_device_count (bindings_dp_common.cu:38-45)."""
    return _capi.load().tplhip_device_count()


def cuda_set_device(idx):
    """tplcpp.cuda_set_device (bindings_dp_common.cu:47-49): the device of every planner / environment created
    afterwards without an explicit ``device=``.  Unlike cudaSetDevice's ignored status, a bad index raises."""
    check(_capi.load().tplhip_set_device(int(idx)))
