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
# column subsets of the reference's array conversions (bindings_dp_lat_lon.cu): a state / a state(ts) or lerp(ts)
# row has 7 columns, a trajectory's as_numpy() prepends t and appends dddl (:56-70, :132-165, :166-184); a
# Cartesian state has 8 columns, a Cartesian lerp(distances) row drops t (:32-44, :83-116)
_STATE_ROW = ("s", "ds", "dds", "ddds", "l", "dl", "ddl")
_STATE_TRAJ_ROW = ("t", "s", "ds", "dds", "ddds", "l", "dl", "ddl", "dddl")
_CART_ROW = ("t", "distance", "x", "y", "v", "a", "heading", "k")
_CART_LERP_ROW = ("distance", "x", "y", "v", "a", "heading", "k")


class DynProgLatLonPlannerParams(_StructView):
    """tplcpp.DynProgLatLonPlannerParams (bindings_dp_lat_lon.cu:188-278)."""
    _ctype = TplLatLonParams
    # the order of the reference's __slots__ list (bindings_dp_lat_lon.cu:233-277), which is not the struct's
    _slot_names = (
        "s_min", "s_max", "ds_min", "ds_max", "l_min", "l_max", "dds_min", "dds_max", "dl_min", "dl_max",
        "t_steps", "s_steps", "ds_steps", "l_steps", "dt", "dt_smooth_traj", "dt_start", "dds_start", "w_dds_start",
        "angle_start", "w_angle_start", "l_trg", "w_progress", "w_dds", "w_ddds", "w_l", "w_dl", "w_ddl",
        "w_xing_slow", "w_safety_dist", "w_lat_dist", "d_lat_comf", "slope_abs_max", "time_gap", "gap_min",
        "t_st_min", "t_st_max", "s_st", "w_spatio_temporal", "width_veh", "length_veh")

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
        return np.array([getattr(self._c, f) for f in _STATE_ROW], dtype=np.float64)


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
        return np.array([getattr(self._c, f) for f in _CART_ROW], dtype=np.float64)


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
        return self._columns(self._traj_row)

    def _columns(self, names):
        """The named members of every state as float64[n, len(names)], read through a record view of the C array."""
        arr = self._as_array()
        if not len(arr):
            return np.empty((0, len(names)), dtype=np.float64)
        rec = np.frombuffer(arr, dtype=np.dtype(self._ctype), count=len(arr))
        return np.stack([rec[f].astype(np.float64) for f in names], axis=1)

    def as_numpy_full(self):
        """Every member of every state (not in the reference: its as_numpy() drops cost / constr / flags)."""
        return self._columns(self._fields)

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
        res = np.empty((len(ts), len(self._many_row)), dtype=np.float64)
        for i, t in enumerate(ts):
            check(fn(arr, len(arr), float(t), C.byref(out)))
            res[i] = [getattr(out, f) for f in self._many_row]
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
    _traj_row = _STATE_TRAJ_ROW
    _many_row = _STATE_ROW

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
    _traj_row = _CART_ROW
    _many_row = _CART_LERP_ROW

    def lerp(self, t):
        fn = _capi.load().tplhip_lat_lon_cart_lerp
        return self._one(fn, t) if np.ndim(t) == 0 else self._many(fn, t)


class LatDistances:
    """tplcpp.LatDistances (bindings_dp_lat_lon.cu:14-17, lat_lon_planner.cuh:123-126)."""
    __slots__ = ("d_left", "d_right")

    def __init__(self, d_left=0.0, d_right=0.0):
        self.d_left = d_left
        self.d_right = d_right


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
        self._device = int(device)
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

    def set_filter_rule(self, weight_quant=1, lerp_form=0, blend_order=0):
        """The texture filter's rules for this planner's filtered fetches (value tables in the forward pass, distance map,
        reference line); ``(1, 0, 0)`` is the default.  See ``tplhip_lat_lon_set_filter_rule`` in include/tplhip.h."""
        check(self._lib.tplhip_lat_lon_set_filter_rule(self._h, int(weight_quant), int(lerp_form), int(blend_order)))

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
        # latDists: the reference resizes it to the length of the PREVIOUS smoothed trajectory and
        # launches getLatDistances over it, a kernel whose body is empty (lat_lon_planner.cu:414-425,
        # 729-747) -- the entries are the default-constructed zeros
        self.lat_dists = [LatDistances() for _ in range(len(self.traj_smooth.states))]
        self.traj_smooth._take(out, n.value)

    def _ref_line_env(self, ref_line):
        """The environment whose reference line the C-ABI converts on.  Accepts what the reference's
        callers pass -- ``cpp_env.ref_line`` (dp_lat_lon_planner.py:148) -- the environment itself, or
        a free-standing RefLine (a private environment is set up for it once and cached)."""
        from .dyn_prog import DynProgEnvironment, RefLine
        if isinstance(ref_line, DynProgEnvironment):
            return ref_line
        if not isinstance(ref_line, RefLine):
            raise TypeError(f"expected a RefLine or a DynProgEnvironment, got {type(ref_line).__name__}")
        owner = ref_line._owner() if getattr(ref_line, "_owner", None) is not None else None
        if owner is not None and owner.ref_line is ref_line:
            return owner
        cached = getattr(self, "_own_env", None)
        if cached is None or cached[0] is not ref_line:
            env = DynProgEnvironment(self._device)
            pts = np.array(ref_line.as_array(), dtype=np.float64, copy=True)
            pts[:, 0] += ref_line.x_offset
            pts[:, 1] += ref_line.y_offset
            env.set_ref_line(pts, ref_line.step_size)
            cached = self._own_env = (ref_line, env)
        return cached[1]

    def frenet_to_cartesian(self, traj, ref_line):
        """bindings_dp_lat_lon.cu:297-303: ``ref_line`` is the RefLine (or the environment holding it)."""
        env = self._ref_line_env(ref_line)
        arr = traj._as_array()
        out = (TplLatLonCartState * len(arr))()
        check(self._lib.tplhip_lat_lon_frenet_to_cartesian(env._h, arr, len(arr), out))
        res = LatLonCartTraj()
        res._take(out)
        return res

    def update_traj_cart(self, ref_line):
        self.traj_dp_cart = self.frenet_to_cartesian(self.traj_dp, ref_line)
        self.traj_smooth_cart = self.frenet_to_cartesian(self.traj_smooth, ref_line)

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
