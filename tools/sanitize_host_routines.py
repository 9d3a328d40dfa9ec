#!/usr/bin/env python3
"""Random inputs through the HOST routines of libtplhip.so (csrc/planning.hip: projection, resampling, ramps, lerp,
reference-line and polygon helpers, prediction geometry, path smoother) in a host-only AddressSanitizer +
UndefinedBehaviorSanitizer build.  CPU only (GPU sanitizers are not available on the pool):

    bash tools/sanitize_host_routines.sh
"""
import ctypes as C, numpy as np, sys
import os
L = C.CDLL(os.environ.get("TPL_SAN_LIB", "/tmp/tpl_san/libplan_san.so"))
dp = C.POINTER(C.c_double)
def P(a): return a.ctypes.data_as(dp)
rng = np.random.default_rng(0)
# gen_prediction_geometry
L.tplhip_gen_prediction_geometry.argtypes = [dp, C.c_int, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_double, C.c_double, C.c_double, dp, C.c_int, C.POINTER(C.c_int)]
for it in range(200):
    n = int(rng.integers(2, 60)); path = np.cumsum(rng.normal(0.5, 0.2, (n, 2)), axis=0)
    k = int(rng.integers(2, 8)); pred = np.zeros((k, 5)); pred[:, 0] = np.sort(rng.uniform(0, 5, k)); pred[:, 1:3] = path[rng.integers(0, n, k)] + rng.normal(0, 1, (k, 2)); pred[:, 3] = rng.uniform(-4, 4, k); pred[:, 4] = rng.uniform(-3, 10)
    hull = path[rng.integers(0, n)] + rng.normal(0, 2, (int(rng.integers(1, 6)), 2))
    ts = np.sort(rng.uniform(0, 6, int(rng.integers(1, 7))))
    cnt = C.c_int()
    assert L.tplhip_gen_prediction_geometry(P(pred), k, P(hull), len(hull), P(path), n, P(ts), len(ts), 5.0, 0.05, 0.5, None, 0, C.byref(cnt)) == 0
    out = np.empty((max(cnt.value, 1), 3))
    assert L.tplhip_gen_prediction_geometry(P(pred), k, P(hull), len(hull), P(path), n, P(ts), len(ts), 5.0, 0.05, 0.5, P(out), cnt.value, C.byref(cnt)) == 0
# polygons, hull, smoother, lerp, ref line
L.tplhip_convex_hull.argtypes = [dp, C.c_int, dp, C.POINTER(C.c_int)]
L.tplhip_smooth_path.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp]
L.tplhip_lerp.argtypes = [dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp]
L.tplhip_point_in_polygon.argtypes = [dp, C.c_int, dp, C.c_int, C.POINTER(C.c_uint8)]
L.tplhip_intersect_polygons.argtypes = [dp, C.c_int, dp, C.c_int, C.POINTER(C.c_int)]
L.tplhip_ref_line_lerp.argtypes = [dp, C.c_int, C.c_float, C.c_float, dp]
for it in range(200):
    n = int(rng.integers(0, 40)); pts = np.round(rng.normal(size=(max(n, 1), 2)), 1)[:n] if n else np.zeros((0, 2))
    out = np.empty((max(n, 1), 2)); cnt = C.c_int()
    assert L.tplhip_convex_hull(P(pts) if n else None, n, P(out), C.byref(cnt)) == 0 and cnt.value <= max(n, 0)
    m = int(rng.integers(2, 50)); path = np.cumsum(rng.normal(0.5, 0.1, (m, 2)), axis=0); o = np.empty_like(path)
    assert L.tplhip_smooth_path(P(path), m, 0.5, 1.0, 1.0, 1.0, int(rng.integers(0, 2)), P(o)) == 0
    xs = np.sort(rng.uniform(0, 10, int(rng.integers(2, 150)))); ys = rng.normal(size=(len(xs), 3)); v = rng.uniform(-2, 12, 7); r = np.empty((7, 3))
    assert L.tplhip_lerp(P(v), 7, P(xs), len(xs), P(ys), 3, int(rng.integers(0, 2)), int(rng.integers(0, 2)), P(r)) == 0
    poly = rng.normal(size=(int(rng.integers(1, 9)), 2)); q = rng.normal(size=(5, 2)); flags = np.zeros(5, np.uint8)
    assert L.tplhip_point_in_polygon(P(q), 5, P(poly), len(poly), flags.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
    hit = C.c_int(); assert L.tplhip_intersect_polygons(P(poly), len(poly), P(q), 5, C.byref(hit)) == 0
    rl = rng.normal(size=(int(rng.integers(1, 30)), 9)); o9 = np.empty(9)
    assert L.tplhip_ref_line_lerp(P(rl), len(rl), 0.5, float(rng.uniform(-5, 30)), P(o9)) == 0
print("sanitized run finished")
L.tplhip_path_resample.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, dp, C.POINTER(C.c_int)]
L.tplhip_rampify_velocity_profile.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, dp]
L.tplhip_rampify_lateral_profile.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, dp, C.c_int, C.c_int, C.c_double, dp, dp, dp]
class Proj(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("alpha", C.c_double), ("index", C.c_int64), ("point", C.c_double * 2),
                ("distance", C.c_double), ("arc_len", C.c_double), ("angle", C.c_double), ("tangent", C.c_double * 2), ("in_bounds", C.c_int)]
L.tplhip_path_project.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.POINTER(Proj)]
for it in range(300):
    n = int(rng.integers(2, 80)); stride = int(rng.choice([2, 6])); path = np.zeros((n, stride)); path[:, :2] = np.cumsum(rng.normal(0.4, 0.3, (n, 2)), axis=0)
    if rng.random() < 0.3: path[n // 2] = path[n // 2 - 1]          # duplicate point
    pos = rng.normal(5, 8, (4, 2)); pr = (Proj * 4)()
    L.tplhip_path_project(P(path), stride, n, int(rng.integers(0, 2)), P(pos), 4, pr)
    steps = int(rng.integers(0, 120)); out = np.zeros((max(steps, 1), 5)); cnt = C.c_int()
    L.tplhip_path_resample(P(path), stride, n, float(rng.uniform(0.1, 2.0)), steps, int(rng.integers(-3, n + 3)), int(rng.integers(0, 2)), P(out), C.byref(cnt))
    assert 0 <= cnt.value <= max(steps, 1)
    m = int(rng.integers(1, 200)); lim = np.clip(rng.normal(8, 4, m), 0, None); o2 = np.zeros((m, 2))
    assert L.tplhip_rampify_velocity_profile(int(rng.integers(0, 2)), 5.0, int(rng.integers(0, 2)), 0.3, P(lim), m, -2.5, 2.0, -1.5, 1.5, 1.0, 0.5, P(o2)) == 0
    h = int(rng.integers(1, m + 1)); pv = np.zeros((m, 6)); pv[:, 5] = lim; lo = rng.normal(-2, 1, m); up = lo + rng.uniform(0.5, 5, m); o1 = np.zeros(m)
    assert L.tplhip_rampify_lateral_profile(0.5, h, 20.0, 0.3, P(pv[:, 5:].copy().ravel()) if False else pv[:, 5:].ctypes.data_as(dp), 6, m, 1.0, P(lo), P(up), P(o1)) == 0
print("second sanitized run finished")
