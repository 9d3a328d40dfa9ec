"""The slice of ``tpl.util`` / ``tplcpp`` / ``tpl.environment.map_module`` / ``objtoolbox`` the path/velocity
glue calls: projection on a line strip, equidistant resampling, angle helpers, velocity-limit helpers, attribute
merge.  ``project`` and ``resample`` are host routines of libtplhip.so (the reference keeps them in its C++
extension too, library/src/utils.cpp:257-548); the rest is numpy."""
import ctypes as C

import numpy as np

from .. import _capi
from .._capi import TplProjection

_dp = C.POINTER(C.c_double)


class Projection:
    """tplcpp.Projection (library/include/tplcpp/utils.hpp:240-256); attributes are writable like the binding's."""
    __slots__ = ("start", "end", "alpha", "index", "point", "distance", "arc_len", "angle", "tangent", "in_bounds")

    def __init__(self, c=None):
        c = c or TplProjection()
        self.start, self.end, self.alpha, self.index = int(c.start), int(c.end), float(c.alpha), int(c.index)
        self.point = np.array([c.point[0], c.point[1]])
        self.distance, self.arc_len, self.angle = float(c.distance), float(c.arc_len), float(c.angle)
        self.tangent = np.array([c.tangent[0], c.tangent[1]])
        self.in_bounds = bool(c.in_bounds)


def project(points, position, closed=False):
    """tplcpp.project (utils.cpp:257-408, binding :895-925): one position -> Projection, a list / array of
    positions -> list of Projections."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] < 2:
        raise ValueError(f"array has incorrect shape, expected: (-1, 2), got: {pts.shape}")
    pos = np.ascontiguousarray(position, dtype=np.float64)
    single = pos.ndim == 1
    pos2 = pos.reshape(-1, 2)
    out = (TplProjection * len(pos2))()
    _capi.check(_capi.load().tplhip_path_project(pts.ctypes.data_as(_dp), pts.shape[1], pts.shape[0], 1 if closed else 0,
                                                  pos2.ctypes.data_as(_dp), len(pos2), out))
    res = [Projection(c) for c in out]
    return res[0] if single else res


def resample(points, sampling_dist, steps, start_index=0, closed=False):
    """tplcpp.resample (utils.cpp:410-548): float64[steps, 5] = (x, y, alpha, prev index, next index);
    RuntimeError where the reference throws."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    out = np.zeros((max(int(steps), 1), 5))
    n = C.c_int()
    _capi.check(_capi.load().tplhip_path_resample(pts.ctypes.data_as(_dp), pts.shape[1], pts.shape[0], float(sampling_dist),
                                                   int(steps), int(start_index), 1 if closed else 0, out.ctypes.data_as(_dp),
                                                   C.byref(n)))
    return out[:n.value]


def gen_prediction_geometry(pred, hull, path, ts=0, station_step_size=5.0, expansion_rate=0.0, sweep_length=0.0):
    """tplcpp.gen_prediction_geometry (utils.cpp:576-693, binding :1007-1048): the triangle soup float64[n, 3] =
    (x, y, t) that DpEnv.update hands to DynProgEnvironment.insert_geometry (dp_env.py:157-168).  ``pred`` is
    float64[k, 5] = (t, x, y, heading, v), ``hull`` float64[m, 2], ``path`` float64[p, 2], ``ts`` the slice times."""
    arrs = []
    for a, shape, name in ((pred, (-1, 5), "pred"), (hull, (-1, 2), "hull"), (path, (-1, 2), "path"), (ts, (-1,), "ts")):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.ndim != len(shape) or (len(shape) == 2 and a.shape[1] != shape[1]):
            raise ValueError(f"array has incorrect shape, expected: {shape}, got: {a.shape}")
        arrs.append(a)
    pred, hull, path, ts = arrs
    lib = _capi.load()
    n = C.c_int()
    args = (pred.ctypes.data_as(_dp), len(pred), hull.ctypes.data_as(_dp), len(hull), path.ctypes.data_as(_dp), len(path),
            ts.ctypes.data_as(_dp), len(ts), float(station_step_size), float(expansion_rate), float(sweep_length))
    _capi.check(lib.tplhip_gen_prediction_geometry(*args, None, 0, C.byref(n)))
    out = np.empty((n.value, 3))
    if n.value:
        _capi.check(lib.tplhip_gen_prediction_geometry(*args, out.ctypes.data_as(_dp), n.value, C.byref(n)))
    return out


def _xy(a, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError(f"array has incorrect shape, expected: (-1, 2), got: {a.shape}")
    return a


def point_in_polygon(points, polygon):
    """tplcpp.point_in_polygon (utils.cpp:25-65, binding :780-804): one point -> bool, an array of points -> bool array."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    if not ((pts.ndim == 1 and pts.shape == (2,)) or (pts.ndim == 2 and pts.shape[1] == 2)):
        raise ValueError(f"array has incorrect shape, expected: (2,) or (-1, 2), got: {pts.shape}")
    poly = _xy(polygon, "polygon")
    p2 = pts.reshape(-1, 2)
    out = np.zeros(len(p2), dtype=np.uint8)
    _capi.check(_capi.load().tplhip_point_in_polygon(p2.ctypes.data_as(_dp), len(p2), poly.ctypes.data_as(_dp), len(poly),
                                                      out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return bool(out[0]) if pts.ndim == 1 else out.astype(bool)


def intersect_polygons(poly0, poly1):
    """tplcpp.intersect_polygons (utils.cpp:67-158): do two convex polygons overlap."""
    a, b = _xy(poly0, "poly0"), _xy(poly1, "poly1")
    out = C.c_int()
    _capi.check(_capi.load().tplhip_intersect_polygons(a.ctypes.data_as(_dp), len(a), b.ctypes.data_as(_dp), len(b), C.byref(out)))
    return bool(out.value)


def convex_hull(points):
    """tplcpp.convex_hull (utils.cpp:160-255): float64[k, 2], counter-clockwise from the lowest point."""
    pts = _xy(points, "points")
    out = np.empty((max(len(pts), 1), 2))
    n = C.c_int()
    _capi.check(_capi.load().tplhip_convex_hull(pts.ctypes.data_as(_dp), len(pts), out.ctypes.data_as(_dp), C.byref(n)))
    return out[:n.value].copy()


def smooth_path(path, ds, w_v=1.0, w_a=1.0, w_j=1.0, closed=False):
    """tplcpp.smooth_path (utils.cpp:693-776): the path through an LQR smoother with jerk inputs."""
    pts = _xy(path, "path")
    out = np.empty_like(pts)
    _capi.check(_capi.load().tplhip_smooth_path(pts.ctypes.data_as(_dp), len(pts), float(ds), float(w_v), float(w_a),
                                                 float(w_j), 1 if closed else 0, out.ctypes.data_as(_dp)))
    return out


def lerp_cpp(vals, xs, ys, angle=False, clip_alpha=True):
    """tplcpp.lerp (utils.cpp:931-990): ``ys`` (1-D or [m, d]) sampled at the sorted ``xs``, evaluated at the 1-D
    ``vals`` -> float64[n] or float64[n, d]."""
    v = np.ascontiguousarray(vals, dtype=np.float64)
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    if v.ndim != 1 or xs.ndim != 1:
        raise ValueError(f"array has incorrect shape, expected: (-1,), got: {v.shape if v.ndim != 1 else xs.shape}")
    if len(v) == 0:
        return np.zeros((0,))
    d = 1 if ys.ndim == 1 else ys.shape[1]
    out = np.empty((len(v), d) if ys.ndim > 1 else (len(v),))
    _capi.check(_capi.load().tplhip_lerp(v.ctypes.data_as(_dp), len(v), xs.ctypes.data_as(_dp), len(xs),
                                         ys.ctypes.data_as(_dp), d, 1 if angle else 0, 1 if clip_alpha else 0,
                                         out.ctypes.data_as(_dp)))
    return out


def lerp(vals, xs, ys, angle=False, clip_alpha=True):
    """tpl.util.lerp (util.py:103-108): a number -> one row, an array -> the squeezed result."""
    if np.ndim(vals) == 0:
        return lerp_cpp([vals], xs, ys, angle, clip_alpha)[0]
    return lerp_cpp(vals, xs, ys, angle, clip_alpha).squeeze()


def curv_to_vel_profile(k, lim_v, a_lat_max, k_eps=1e-6):
    """tpl.environment.map_module.curv_to_vel_profile (map_module.py:534-545): the speed limit capped by the lateral
    acceleration bound, sqrt(a_lat_max / |k|), wherever the curvature is not negligible."""
    abs_k = np.abs(np.asarray(k, dtype=np.float64))
    lim_v = np.asarray(lim_v, dtype=np.float64)
    with np.errstate(divide="ignore"):
        capped = np.minimum(lim_v, np.sqrt(a_lat_max / abs_k))
    return np.where(abs_k > k_eps, capped, lim_v)


def normalize_angle(a):
    """tpl.util.normalize_angle (util.py:92-100): into (-pi, pi]."""
    a = a % (np.pi * 2)
    a = (a + np.pi * 2) % (np.pi * 2)
    return np.where(a > np.pi, a - np.pi * 2, a) if isinstance(a, np.ndarray) else (a - np.pi * 2 if a > np.pi else a)


def short_angle_dist(x, y):
    """tpl.util.short_angle_dist (util.py:70-89), elementwise."""
    x = normalize_angle(np.asarray(x, dtype=np.float64))
    y = normalize_angle(np.asarray(y, dtype=np.float64))
    a0, a1, a2 = y - x, y - x + 2 * np.pi, y - x - 2 * np.pi
    a = np.where(np.abs(a1) < np.abs(a0), a1, a0)
    return np.where(np.abs(a2) < np.abs(a), a2, a)


def resample_path(path, step_size, steps, start_index=0, zero_vel_at_end=False, closed=False):
    """tpl.util.resample_path + interp_resampled_path (util.py:134-191): a path array (x, y, orientation, s, k, v)
    in equidistant steps; None where the resampling fails, like the reference."""
    path = np.array(path, dtype=np.float64)
    try:
        rsi = resample(path[:, :2], step_size, steps, start_index, closed)
    except RuntimeError:
        return None
    rs = np.zeros((steps, 6))
    rs[:, :2] = rsi[:, :2]
    prevs = path[rsi[:, 3].astype(int)]
    nexts = path[rsi[:, 4].astype(int)]
    t = rsi[:, 2]
    extra = (rsi[:, 4] == len(path) - 1) & (t > 1.0) if not closed else np.zeros(len(rsi), dtype=bool)
    rs[:, 2] = np.where(extra, nexts[:, 2], prevs[:, 2] + t * short_angle_dist(prevs[:, 2], nexts[:, 2]))
    rs[:, 3] = step_size * np.arange(len(rsi))
    rs[:, 5] = np.where(extra, 0.0 if zero_vel_at_end else nexts[:, 5], (1.0 - t) * prevs[:, 5] + t * nexts[:, 5])
    rs[:-1, 4] = 2 * np.sin(short_angle_dist(rs[:-1, 2], rs[1:, 2]) / 2) / step_size
    if closed:
        gap = np.linalg.norm(rs[0, :2] - rs[-1, :2])
        rs[-1, 4] = rs[-2, 4] if gap == 0.0 else 2 * np.sin(short_angle_dist(rs[-1, 2], rs[0, 2]) / 2) / gap
    else:
        rs[-1, 4] = rs[-2, 4]
    return rs


def add_vel_constraint(lim_v, index, max_vel=0.0, length=10, shift=0):
    """tpl.environment.map_module.add_vel_constraint (map_module.py:560-570), in place."""
    i = int(index + shift)
    l = min(len(lim_v) - i, int(length))
    j = max(0, i + l)
    i = max(0, i)
    lim_v[i:j] = np.minimum(lim_v[i:j], max_vel)


def zero_after_first_zero(vel_profile):
    """tpl.environment.map_module.zero_after_first_zero (map_module.py:549-557)."""
    v = np.asarray(vel_profile)
    zeros = np.flatnonzero(v == 0)
    out = np.zeros_like(v, dtype=np.float64)
    out[:zeros[0] if len(zeros) else len(v)] = 1.0
    return out


def merge(dst, src):
    """objtoolbox.merge as the glue uses it (path_optim.py:133, velocity_optim.py:162): copy every public
    attribute of ``src`` that ``dst`` also has (``dst``: a solver's Params -- ``__slots__`` -- or a plain object)."""
    names = getattr(dst, "__slots__", None) or [k for k in vars(dst) if not k.startswith("_")]
    src_names = (getattr(src, "__slots__", None) or [k for k in vars(src) if not k.startswith("_")])
    for k in src_names:
        if k in names:
            setattr(dst, k, getattr(src, k))
