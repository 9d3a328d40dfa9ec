"""Python surface of the lateral quintic-lattice planner: the names, fields and
methods of tplcpp's PolyLatPlanner family (library/src/python/bindings_dp_poly_lat.cu), computed
by tpl_amd/csrc/poly_lat.hip through the C-ABI of include/tplhip.h."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import TplPolyLatParams, TplPolyLatTrajPoint, TplPolyQuintic
from .dyn_prog import _StructView, check

_FIELDS12 = ("t", "l", "dl", "ddl", "dddl", "s", "v", "x", "y", "heading", "distance", "k")


class PolyLatPlannerParams(_StructView):
    """tplcpp.PolyLatPlannerParams (bindings_dp_poly_lat.cu:102-161)."""
    _ctype = TplPolyLatParams
    _slot_names = (
        "l_min", "l_max", "s_min", "s_max", "s_steps", "l_dst_min", "l_dst_max", "s_dst_min", "s_dst_max",
        "l_dst_steps", "s_dst_steps", "l_trg", "w_l", "w_k", "w_dl", "w_ddl", "w_dddl", "w_right", "w_len",
        "k_abs_max", "a_lat_abs_max", "width_veh", "length_veh")

    def _set_defaults(self):
        _capi.load().tplhip_poly_lat_params_default(C.byref(self._c))

    def copy(self):
        return self.__deepcopy__({})


class PolyLatTrajPoint(_StructView):
    """tplcpp.PolyLatTrajPoint (bindings_dp_poly_lat.cu:9-45)."""
    _ctype = TplPolyLatTrajPoint
    _slot_names = _FIELDS12

    def _set_defaults(self):
        pass                                    # all members default to 0 (poly_lat_planner.cuh:61-76)

    @classmethod
    def _from_c(cls, c):
        pt = cls()
        C.memmove(C.byref(pt._c), C.byref(c), C.sizeof(TplPolyLatTrajPoint))
        return pt

    def copy(self):
        return self.__deepcopy__({})

    def as_numpy(self):
        return np.array([getattr(self._c, f) for f in _FIELDS12], dtype=np.float64)


class PolyQuintic:
    """tplcpp.PolyQuintic as far as tpl's planners read it (x0_, x1_, f, df, ddf, dddf)."""

    def __init__(self, *args):
        self._c = TplPolyQuintic()
        if args:
            _capi.load().tplhip_poly_quintic_init(C.byref(self._c), *[float(a) for a in args])

    x0_ = property(lambda self: self._c.x0)
    x1_ = property(lambda self: self._c.x1)
    d = property(lambda self: self._c.d)

    def _eval(self, order, x):
        return _capi.load().tplhip_poly_quintic_eval(C.byref(self._c), order, float(x))

    def f(self, x):
        return self._eval(0, x)

    def df(self, x):
        return self._eval(1, x)

    def ddf(self, x):
        return self._eval(2, x)

    def dddf(self, x):
        return self._eval(3, x)


_POINT_DTYPE = np.dtype(TplPolyLatTrajPoint)
assert _POINT_DTYPE.itemsize == C.sizeof(TplPolyLatTrajPoint)


class PolyLatTraj:
    """tplcpp.PolyLatTraj (bindings_dp_poly_lat.cu:47-100)."""

    def __init__(self):
        self._points = []
        self._arr = None                       # C array of a fresh result; `points` not built yet
        self.poly = PolyQuintic()
        self.cost = 0.0

    # `points` is the reference's list of PolyLatTrajPoint.  A planner result arrives as one C array
    # and the list (one Python object per station) is only built when somebody asks for it: lerp,
    # as_numpy and the planners' own use go through the array.
    @property
    def points(self):
        if self._points is None:
            self._points = [PolyLatTrajPoint._from_c(c) for c in self._arr]
            self._arr = None
        return self._points

    @points.setter
    def points(self, value):
        self._points = value
        self._arr = None

    def __len__(self):
        return len(self._arr) if self._points is None else len(self._points)

    def _as_array(self):
        if self._points is None:
            arr = (TplPolyLatTrajPoint * len(self._arr))()
            C.memmove(arr, self._arr, C.sizeof(arr))
            return arr
        arr = (TplPolyLatTrajPoint * len(self._points))()
        for i, p in enumerate(self._points):
            C.memmove(C.byref(arr[i]), C.byref(p._c), C.sizeof(TplPolyLatTrajPoint))
        return arr

    def _take(self, arr):
        self._arr = arr
        self._points = None

    def update_time_dist_curv(self):
        arr = self._as_array()
        check(_capi.load().tplhip_poly_lat_traj_update(arr, len(arr)))
        self._take(arr)

    @staticmethod
    def _view(arr):
        """The C array as a numpy record array (same memory)."""
        return np.frombuffer(arr, dtype=_POINT_DTYPE, count=len(arr))

    def _stations(self, arr):
        return self._view(arr)["s"] if len(arr) else np.zeros(0, np.float32)

    def insert_after_station(self, s, other):
        # PolyLatTraj::insertAfterStation, poly_lat_planner.cu:283-298: everything from the first point with
        # station >= s on is replaced by the other trajectory's points; done on the C arrays (no per-point objects)
        mine, theirs = self._as_array(), other._as_array()
        hit = np.flatnonzero(self._stations(mine) >= s)
        keep = int(hit[0]) if len(hit) else len(mine)
        arr = (TplPolyLatTrajPoint * (keep + len(theirs)))()
        sz = C.sizeof(TplPolyLatTrajPoint)
        C.memmove(arr, mine, keep * sz)
        if len(theirs):
            C.memmove(C.byref(arr, keep * sz), theirs, len(theirs) * sz)
        check(_capi.load().tplhip_poly_lat_traj_update(arr, len(arr)))
        self._take(arr)

    def shift_stations(self, shift):
        """``for p in traj.points: p.s -= shift`` followed by dropping the points with p.s < 0 -- what
        PolyLatDpLonPlanner.shift_trajectory does through the point list (poly_lat_dp_lon_planner.py:305-307) -- on
        the C array."""
        arr = self._as_array()
        if not len(arr):
            return
        a = self._view(arr)
        a["s"] = (a["s"].astype(np.float64) - shift).astype(np.float32)
        kept = np.ascontiguousarray(a[a["s"] >= 0.0])
        out = (TplPolyLatTrajPoint * len(kept))()
        C.memmove(out, kept.ctypes.data, len(kept) * C.sizeof(TplPolyLatTrajPoint))
        self._take(out)

    def lerp(self, distance):
        arr = self._as_array()
        lib = _capi.load()
        if np.ndim(distance) == 0:
            out = TplPolyLatTrajPoint()
            check(lib.tplhip_poly_lat_traj_lerp(arr, len(arr), float(distance), C.byref(out)))
            return PolyLatTrajPoint._from_c(out)
        distance = np.ascontiguousarray(distance, dtype=np.float64)
        if distance.ndim != 1:
            raise ValueError(f"array has incorrect shape, expected: (-1,), got: {distance.shape}")
        res = np.empty((len(distance), 12), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        check(lib.tplhip_poly_lat_traj_lerp_rows(arr, len(arr), distance.ctypes.data_as(dp), len(distance),
                                                 res.ctypes.data_as(dp)))
        return res

    def as_numpy(self):
        arr = self._as_array()
        res = np.empty((len(arr), 12), dtype=np.float64)
        check(_capi.load().tplhip_poly_lat_traj_rows(arr, len(arr), res.ctypes.data_as(C.POINTER(C.c_double))))
        return res

    def copy(self):
        other = PolyLatTraj()
        if self._points is None:
            other._take(self._as_array())
        else:
            other.points = [p.copy() for p in self._points]
        C.memmove(C.byref(other.poly._c), C.byref(self.poly._c), C.sizeof(TplPolyQuintic))
        other.cost = self.cost
        return other


class PolyLatPlanner:
    """tplcpp.PolyLatPlanner (bindings_dp_poly_lat.cu:163-172)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        self._lib = _capi.load()
        check(self._lib.tplhip_poly_lat_create(int(device), C.byref(self._h)))
        self.params = PolyLatPlannerParams()
        self.best = (0, 0)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.tplhip_poly_lat_destroy(h)
            self._h = None

    def reinit_buffers(self, params):
        check(self._lib.tplhip_poly_lat_reinit(self._h, C.byref(params._c)))
        self.params = params.__deepcopy__({})

    def update(self, start, env):
        n = self.params.s_steps
        pts = (TplPolyLatTrajPoint * n)()
        traj = PolyLatTraj()
        bl, bs = C.c_int(), C.c_int()
        check(self._lib.tplhip_poly_lat_update(self._h, C.byref(start._c), env._h, pts, C.byref(traj.poly._c),
                                               C.byref(bl), C.byref(bs)))
        traj._take(pts)
        self.best = (bl.value, bs.value)
        return traj

    def nodes(self):
        """float32[l_dst_steps, s_dst_steps, s_steps, 4]: the lattice of the last update (debug / parity)."""
        p = self.params
        out = np.empty((p.l_dst_steps, p.s_dst_steps, p.s_steps, 4), dtype=np.float32)
        check(self._lib.tplhip_poly_lat_copy_nodes(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out
