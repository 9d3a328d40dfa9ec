"""CPU restatement (numpy / plain Python) of the reference's prediction-geometry generator and the two routines it
calls.  TEST INFRASTRUCTURE: only tests/ may import this module; the product's implementation is the host C++ of
tpl_amd/csrc/planning.hip, written separately from the same source.

  project(points, position)      library/src/utils.cpp:257-408 (open line strip only: `closed` is false on this path)
  interp_prediction(pred, t)     library/src/utils.cpp:560-574, InterpVars library/include/tplcpp/utils.hpp:58-86
  gen_prediction_geometry(...)   library/src/utils.cpp:576-693

Parity unpinned: utils.cpp needs Eigen and pybind11 to compile, which this image lacks, and the reference's tests
hold no vector for these routines; the checks are hand-derived cases (tests/test_prediction_geometry.py).
"""
import math

import numpy as np


def short_angle_dist(x, y):
    """utils.hpp:21-53"""
    def norm(a):
        a = math.fmod(a, 2 * math.pi)
        a = math.fmod(a + 2 * math.pi, 2 * math.pi)
        return a - 2 * math.pi if a > math.pi else a
    x, y = norm(x), norm(y)
    a0, a1, a2 = y - x, y - x + 2 * math.pi, y - x - 2 * math.pi
    if abs(a1) < abs(a0):
        a0 = a1
    if abs(a2) < abs(a0):
        a0 = a2
    return a0


def project(points, position):
    """-> dict(in_bounds, arc_len, distance, index, alpha, point) of the closest point of the open strip."""
    pts = np.asarray(points, dtype=np.float64)
    pos = np.asarray(position, dtype=np.float64)
    n = len(pts)
    best, off, arc = math.inf, 0.0, 0.0
    res = dict(in_bounds=False, index=0, alpha=0.0, point=pts[0].copy())
    prev = pts[0]
    for i in range(1, n):
        nxt = pts[i]
        pv, v = pos - prev, nxt - prev
        seg = math.sqrt(v[0] * v[0] + v[1] * v[1])
        q = (pv[0] * v[0] + pv[1] * v[1]) / (v[0] * v[0] + v[1] * v[1])
        in_bounds = True
        if q < 0:
            in_bounds = i != 1
            q, c = 0.0, prev
        elif q > 1:
            in_bounds = i != n - 1
            q, c = 1.0, nxt
        else:
            c = prev + v * q
        d = (pos[0] - c[0]) ** 2 + (pos[1] - c[1]) ** 2
        if d < best:
            best, off = d, arc
            res.update(in_bounds=in_bounds, index=i, alpha=q, point=np.array(c, dtype=np.float64))
        arc += seg
        prev = nxt
    dist = math.sqrt(best)
    i_start, i_end = max(0, res["index"] - 1), res["index"]
    a, b = pts[i_start], pts[i_end]
    v = b - a
    v = v / math.sqrt(v[0] * v[0] + v[1] * v[1])
    ad = a - res["point"]
    res["arc_len"] = off + math.sqrt(ad[0] * ad[0] + ad[1] * ad[1]) * (-1.0 if res["alpha"] < 0 else 1.0)
    o = res["point"] - pos
    ol = math.sqrt(o[0] * o[0] + o[1] * o[1])
    with np.errstate(invalid="ignore", divide="ignore"):
        o = o / ol
    r = np.array([-o[1], o[0]])
    if v[0] * r[0] + v[1] * r[1] <= 0:
        dist = -dist
    res["distance"] = dist
    return res


def interp_prediction(pred, t):
    pred = np.asarray(pred, dtype=np.float64)
    ts = pred[:, 0]
    n = len(pred)
    i_prev = i_next = 0
    if t < ts[0]:
        i_prev, i_next = 0, 1
    elif t >= ts[-1]:
        i_prev, i_next = n - 2, n - 1
    else:
        for i in range(1, n):
            if ts[i - 1] <= t < ts[i]:
                i_prev, i_next = i - 1, i
                break
    a = (t - ts[i_prev]) / (ts[i_next] - ts[i_prev])
    a = max(0.0, min(1.0, a))
    p, q = pred[i_prev], pred[i_next]
    return np.array([t, p[1] * (1.0 - a) + q[1] * a, p[2] * (1.0 - a) + q[2] * a,
                     p[3] + short_angle_dist(p[3], q[3]) * a, p[4] * (1.0 - a) + q[4] * a])


def gen_prediction_geometry(pred, hull, path, ts, station_step_size=5.0, expansion_rate=0.0, sweep_length=0.0):
    pred, hull, path, ts = (np.asarray(a, dtype=np.float64) for a in (pred, hull, path, ts))
    d01 = path[1] - path[0]
    path_step = math.sqrt(d01[0] * d01[0] + d01[1] * d01[1])

    def get_path(s):
        a = s / path_step
        i_prev = max(0, min(len(path) - 2, int(math.floor(a))))
        i_next = max(1, min(len(path) - 1, int(math.ceil(a))))
        a = max(0.0, min(1.0, a - i_prev))
        return path[i_prev] * (1.0 - a) + path[i_next] * a

    s_min, s_max, d_min, d_max = math.inf, -math.inf, math.inf, -math.inf
    for v in hull:
        pr = project(path, v)
        if pr["in_bounds"]:
            s_min, s_max = min(pr["arc_len"], s_min), max(pr["arc_len"], s_max)
            d_min, d_max = min(pr["distance"], d_min), max(pr["distance"], d_max)
    pp0 = interp_prediction(pred, ts[0])
    length = s_max - s_min
    s_mid = project(path, pp0[1:3])["arc_len"]
    vtxs = []
    for k in range(len(ts) - 1):
        t, dt = ts[k], ts[k + 1] - ts[k]
        pp = interp_prediction(pred, t)
        sg = -1.0 if pp[4] < 0.0 else 1.0
        s_start = s_mid - sg * length * 0.5
        s_stop = s_mid + max(sweep_length, dt) * pp[4] + sg * length * 0.5
        x = abs(s_stop - s_start) / station_step_size + 1
        # (int) of a non-finite double is INT_MIN on x86-64 (cvttsd2si): the loop below does not run
        steps = int(x) if math.isfinite(x) and abs(x) < 2 ** 31 else -2 ** 31
        for i in range(steps):
            s = s_start + i * station_step_size * sg
            ds = sg * min(abs(s_stop - s), station_step_size)
            if abs(ds) < 1e-3:
                break
            p0, p1 = get_path(s), get_path(s + ds)
            v = (p1 - p0) * sg
            vl = math.sqrt(v[0] * v[0] + v[1] * v[1])
            if vl < 1e-3:
                break
            ortho = np.array([-v[1], v[0]]) / vl
            lo0, lo1 = p0 + ortho * d_min, p1 + ortho * d_min
            hi0, hi1 = p0 + ortho * d_max, p1 + ortho * d_max
            if i == 0:
                vtxs += [(lo0[0], lo0[1], t), (lo1[0], lo1[1], t), (hi0[0], hi0[1], t),
                         (hi0[0], hi0[1], t), (lo1[0], lo1[1], t), (hi1[0], hi1[1], t)]
            else:
                prev_min, prev_max = vtxs[-2], vtxs[-1]
                vtxs += [prev_min, (lo1[0], lo1[1], t), prev_max, prev_max, (lo1[0], lo1[1], t), (hi1[0], hi1[1], t)]
        length *= 1.0 + expansion_rate
        s_mid += dt * pp[4]
    return np.array(vtxs, dtype=np.float64).reshape(-1, 3)


# ---------------------------------------------------------------------------------------------------------------
# polygon helpers and the path smoother of the same source file (utils.cpp:25-255, 693-776; lqrSmoother utils.hpp:278-326)

def point_in_polygon(x, y, polygon):
    poly = np.asarray(polygon, dtype=np.float64)
    n = len(poly)
    p1x, p1y = poly[0]
    xints, inside = 0.0, False
    for i in range(n + 1):
        p2x, p2y = poly[i % n]
        if min(p1y, p2y) <= y <= max(p1y, p2y) and x <= max(p1x, p2x):
            if p1y != p2y:
                xints = (y - p1y) * (p2x - p1x) / (p2y - p1y) + p1x
            if p1x == p2x or x <= xints:
                inside = not inside
        p1x, p1y = p2x, p2y
    return inside


def intersect_polygons(poly0, poly1):
    a, b = np.asarray(poly0, dtype=np.float64), np.asarray(poly1, dtype=np.float64)
    if len(a) == 0 or len(b) == 0:
        return False
    for p in (a, b):
        for i in range(len(p)):
            e = p[(i + 1) % len(p)] - p[i]
            e = e / math.sqrt(e[0] * e[0] + e[1] * e[1])
            o = np.array([-e[1], e[0]])
            q0, q1 = (a - p[i]) @ o, (b - p[i]) @ o
            if (q0.max() < q1.min() and q0.max() < q1.max()) or (q0.min() > q1.min() and q0.min() > q1.max()):
                return False
    return True


def smooth_path(path, ds, w_v=1.0, w_a=1.0, w_j=1.0, closed=False):
    path = np.asarray(path, dtype=np.float64)
    H = len(path)
    x_ref = np.zeros((H, 6))
    x_ref[:, 0], x_ref[:, 3] = path[:, 0], path[:, 1]
    d = path[1] - path[0]
    d = d / math.sqrt(d[0] * d[0] + d[1] * d[1]) * ds
    x_ref[0, 1], x_ref[0, 4] = d
    if closed:
        x_ref[-1] = x_ref[0]
    A = np.eye(6)
    A[0, 1] = A[1, 2] = A[3, 4] = A[4, 5] = ds
    B = np.zeros((6, 2))
    B[2, 0] = B[5, 1] = ds
    Q = np.diag([1.0, w_v, w_a, 1.0, w_v, w_a])
    R = np.eye(2) * w_j
    Qs = [Q.copy() for _ in range(H)]
    Qs[0] = np.eye(6) * 1.0e6
    Qs[-1] = np.eye(6) * 1.0e6
    k, K = [None] * H, [None] * H
    Vxx = Qs[-1].copy()
    Vx = -Qs[-1] @ x_ref[-1]
    for i in range(H - 2, -1, -1):
        lx = -Qs[i] @ x_ref[i]
        Qx = lx + A.T @ Vx
        Qu = B.T @ Vx
        Qxx = Qs[i] + A.T @ Vxx @ A
        Quu = R + B.T @ Vxx @ B
        Qux = B.T @ Vxx @ A
        W = -np.linalg.inv(Quu)
        k[i], K[i] = W @ Qu, W @ Qux
        Vx = Qx + K[i].T @ Quu @ k[i] + K[i].T @ Qu + Qux.T @ k[i]
        Vxx = Qxx + K[i].T @ Quu @ K[i]
        tmp = K[i].T @ Qux
        Vxx = Vxx + tmp + tmp.T
    xs = np.zeros((H, 6))
    xs[0] = x_ref[0]
    for i in range(H - 1):
        u = K[i] @ xs[i] + k[i]
        xs[i + 1] = A @ xs[i] + B @ u
    return xs[:, [0, 3]]
