"""Python surface of the DP (s, ds, l) lateral/longitudinal planner: the names,
fields and methods of tplcpp's DynProgLatLonPlanner family
(library/src/python/bindings_dp_lat_lon.cu), computed by tpl_amd/csrc/lat_lon.hip."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import TplLatLonCartState, TplLatLonParams, TplLatLonState
from .dyn_prog import _StructView, check

LAT_LON_CONSTR_OCCUPANCY, LAT_LON_CONSTR_VELOCITY, LAT_LON_CONSTR_ANGLE = 1, 2, 4

_STATE_F = ("t", "s", "ds", "dds", "ddds", "l", "dl", "ddl", "dddl", "cost", "constr", "flags_constr")
_CART_F = ("t", "distance", "x", "y", "v", "a", "heading", "k", "cost", "constr", "flags_constr")


class DynProgLatLonPlannerParams(_StructView):
    """tplcpp.DynProgLatLonPlannerParams (bindings_dp_lat_lon.cu:188-278)."""
    _ctype = TplLatLonParams
    _slot_names = tuple(f[0] for f in TplLatLonParams._fields_)

    def _set_defaults(self):
        _capi.load().tplhip_lat_lon_params_default(C.byref(self._c))


class LatLonState(_StructView):
    """tplcpp.LatLonState (bindings_dp_lat_lon.cu:46-71)."""
    _ctype = TplLatLonState
    _slot_names = _STATE_F

    def _set_defaults(self):
        pass

    @classmethod
    def _from_c(cls, c):
        st = cls()
        C.memmove(C.byref(st._c), C.byref(c), C.sizeof(TplLatLonState))
        return st

    def as_numpy(self):
        return np.array([getattr(self._c, f) for f in _STATE_F], dtype=np.float64)


class LatLonCartState(_StructView):
    """tplcpp.LatLonCartState (bindings_dp_lat_lon.cu:19-44); ``constr`` defaults to 1 like the
    reference's ``double constr = true``."""
    _ctype = TplLatLonCartState
    _slot_names = _CART_F

    def _set_defaults(self):
        self._c.constr = 1.0

    @classmethod
    def _from_c(cls, c):
        st = cls()
        C.memmove(C.byref(st._c), C.byref(c), C.sizeof(TplLatLonCartState))
        return st

    def as_numpy(self):
        return np.array([getattr(self._c, f) for f in _CART_F], dtype=np.float64)


class _Traj:
    _state_cls = None
    _ctype = None
    _fields = ()

    def __init__(self):
        self._states = [self._state_cls()]         # std::vector<...> states{1}
        self._arr = None                           # (C array, length) of a fresh result; list not built yet

    # `states` is the reference's list of state objects.  Results of the planner arrive as one C
    # array; the list (one Python object per state) is built when somebody asks for it, while
    # state(t), lerp(t), as_numpy() and the planner's own calls work on the array.
    @property
    def states(self):
        if self._states is None:
            arr, n = self._arr
            self._states = [self._state_cls._from_c(arr[i]) for i in range(n)]
            self._arr = None
        return self._states

    @states.setter
    def states(self, value):
        self._states = value
        self._arr = None

    def __len__(self):
        return self._arr[1] if self._states is None else len(self._states)

    def __getitem__(self, idx):
        return self.states[idx]

    def __setitem__(self, idx, val):
        self.states[idx] = val.__deepcopy__({})

    def _as_array(self):
        if self._states is None:
            src, n = self._arr
            arr = (self._ctype * n)()
            C.memmove(arr, src, C.sizeof(arr))
            return arr
        arr = (self._ctype * len(self._states))()
        for i, s in enumerate(self._states):
            C.memmove(C.byref(arr[i]), C.byref(s._c), C.sizeof(self._ctype))
        return arr

    def _take(self, arr, n=None):
        self._arr = (arr, len(arr) if n is None else n)
        self._states = None

    def as_numpy(self):
        arr = self._as_array()
        res = np.empty((len(arr), len(self._fields)), dtype=np.float64)
        for i, c in enumerate(arr):
            res[i] = [getattr(c, f) for f in self._fields]
        return res

    def copy(self):
        other = type(self)()
        if self._states is None:
            other._take(self._as_array())
        else:
            other.states = [s.__deepcopy__({}) for s in self._states]
        return other

    def _many(self, fn, ts):
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        if ts.ndim != 1:
            raise ValueError(f"array has incorrect shape, expected: (-1,), got: {ts.shape}")
        arr = self._as_array()
        out = self._ctype()
        res = np.empty((len(ts), len(self._fields)), dtype=np.float64)
        for i, t in enumerate(ts):
            check(fn(arr, len(arr), float(t), C.byref(out)))
            res[i] = [getattr(out, f) for f in self._fields]
        return res

    def _one(self, fn, t):
        arr = self._as_array()
        out = self._ctype()
        check(fn(arr, len(arr), float(t), C.byref(out)))
        return self._state_cls._from_c(out)


class LatLonTraj(_Traj):
    """tplcpp.LatLonTraj (bindings_dp_lat_lon.cu:121-186)."""
    _state_cls = LatLonState
    _ctype = TplLatLonState
    _fields = _STATE_F

    def state(self, t):
        fn = _capi.load().tplhip_lat_lon_traj_state
        return self._one(fn, t) if np.ndim(t) == 0 else self._many(fn, t)

    def lerp(self, t):
        fn = _capi.load().tplhip_lat_lon_traj_lerp
        return self._one(fn, t) if np.ndim(t) == 0 else self._many(fn, t)


class LatLonCartTraj(_Traj):
    """tplcpp.LatLonCartTraj (bindings_dp_lat_lon.cu:73-119)."""
    _state_cls = LatLonCartState
    _ctype = TplLatLonCartState
    _fields = _CART_F

    def lerp(self, t):
        fn = _capi.load().tplhip_lat_lon_cart_lerp
        return self._one(fn, t) if np.ndim(t) == 0 else self._many(fn, t)


def lat_lon_dynamics(x0, dds, dl, dt):
    out = TplLatLonState()
    _capi.load().tplhip_lat_lon_dynamics(C.byref(x0._c), float(dds), float(dl), float(dt), C.byref(out))
    return LatLonState._from_c(out)


class DynProgLatLonPlanner:
    """tplcpp.DynProgLatLonPlanner (bindings_dp_lat_lon.cu:280-317)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        self._lib = _capi.load()
        check(self._lib.tplhip_lat_lon_create(int(device), C.byref(self._h)))
        self.params = DynProgLatLonPlannerParams()
        self.traj_dp = LatLonTraj()
        self.traj_smooth = LatLonTraj()
        self.traj_dp_cart = LatLonCartTraj()
        self.traj_smooth_cart = LatLonCartTraj()
        self.lat_dists = []
        self.reinit_buffers(self.params, True)          # lat_lon_planner.cu:488-491

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.tplhip_lat_lon_destroy(h)
            self._h = None

    def reinit_buffers(self, params, force=False):
        full = force or any(getattr(params, f) != getattr(self.params, f)
                            for f in ("t_steps", "s_steps", "ds_steps", "l_steps"))
        check(self._lib.tplhip_lat_lon_reinit(self._h, C.byref(params._c), 1 if force else 0))
        self.params = params.__deepcopy__({})
        if full:                                         # :543-545
            old = self.traj_dp.states
            self.traj_dp.states = (old + [LatLonState() for _ in range(self.params.t_steps)])[:self.params.t_steps]
            self.traj_smooth.states = self.traj_smooth.states[:1]
            self.traj_smooth_cart.states = self.traj_smooth_cart.states[:1]

    def update_traj_dp(self, env):
        T = self.params.t_steps
        cur = self.traj_dp._as_array()
        arr = (TplLatLonState * T)()                                             # states.resize(t_steps)
        C.memmove(arr, cur, min(len(cur), T) * C.sizeof(TplLatLonState))
        check(self._lib.tplhip_lat_lon_update_traj_dp(self._h, env._h, arr))
        self.traj_dp._take(arr)

    def update_traj_smooth(self, env):
        dp = self.traj_dp._as_array()
        x0 = self.traj_smooth._as_array()[0]
        cap = max(2, int(dp[len(dp) - 1].t / self.params.dt_smooth_traj) + 2)
        out = (TplLatLonState * cap)()
        n = C.c_int()
        check(self._lib.tplhip_lat_lon_update_traj_smooth(self._h, dp, len(dp), C.byref(x0), out, cap, C.byref(n)))
        self.traj_smooth._take(out, n.value)

    def frenet_to_cartesian(self, traj, env):
        """(the reference takes the RefLine; here the environment that holds it)"""
        arr = traj._as_array()
        out = (TplLatLonCartState * len(arr))()
        check(self._lib.tplhip_lat_lon_frenet_to_cartesian(env._h, arr, len(arr), out))
        res = LatLonCartTraj()
        res._take(out)
        return res

    def update_traj_cart(self, env):
        self.traj_dp_cart = self.frenet_to_cartesian(self.traj_dp, env)
        self.traj_smooth_cart = self.frenet_to_cartesian(self.traj_smooth, env)

    def reeval_traj(self, traj, env):
        arr = traj._as_array()
        out = (TplLatLonState * len(arr))()
        check(self._lib.tplhip_lat_lon_reeval_traj(self._h, env._h, arr, len(arr), out))
        res = LatLonTraj()
        res._take(out)
        return res

    def query_backward_pass_tex(self, idx_t, s, ds, l):
        out = np.empty(4, dtype=np.float32)
        check(self._lib.tplhip_lat_lon_query_backward_pass_tex(self._h, int(idx_t), float(s), float(ds), float(l),
                                                               out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def copy_backward_pass_tex(self, idx_t):
        p = self.params
        out = np.empty((p.l_steps, p.ds_steps, p.s_steps, 4), dtype=np.float32)
        check(self._lib.tplhip_lat_lon_copy_backward_pass_tex(self._h, int(idx_t), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out
