"""tplcpp's polygon helpers and path smoother (utils.cpp:25-255, 693-776) as host routines of libtplhip.so, against the
numpy restatement in oracle/planning_oracle.py, scipy's hull, and cases worked by hand.  No GPU."""
import importlib.util
import os

import numpy as np
import pytest

import tpl_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("planning_oracle", os.path.join(ROOT, "oracle", "planning_oracle.py"))
po = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(po)

SQUARE = np.array([[0.0, 0.0], [2.0, 0.0], [2.0, 2.0], [0.0, 2.0]])


def test_point_in_polygon_by_hand_and_against_the_restatement():
    assert tpl_amd.point_in_polygon([1.0, 1.0], SQUARE) is True and tpl_amd.point_in_polygon([3.0, 1.0], SQUARE) is False
    got = tpl_amd.point_in_polygon(np.array([[1.0, 1.0], [3.0, 1.0], [-1.0, 1.0], [1.0, 2.5]]), SQUARE)
    assert got.dtype == bool and got.tolist() == [True, False, False, False]
    rng = np.random.default_rng(0)
    ang = np.sort(rng.uniform(0, 2 * np.pi, 9))
    star = np.stack([np.cos(ang), np.sin(ang)], axis=1) * rng.uniform(0.4, 1.5, 9)[:, None]      # non-convex
    pts = rng.uniform(-1.6, 1.6, (400, 2))
    pts[:20] = star[rng.integers(0, 9, 20)]                      # vertices themselves: the source's own edge rule
    want = np.array([po.point_in_polygon(x, y, star) for x, y in pts])
    assert np.array_equal(tpl_amd.point_in_polygon(pts, star), want) and 50 < want.sum() < 350
    with pytest.raises(ValueError, match="expected"):
        tpl_amd.point_in_polygon([1.0, 2.0, 3.0], SQUARE)


def test_intersect_polygons():
    assert tpl_amd.intersect_polygons(SQUARE, SQUARE + 1.5) and not tpl_amd.intersect_polygons(SQUARE, SQUARE + 2.5)
    assert tpl_amd.intersect_polygons(SQUARE, SQUARE + [2.0, 0.0])               # touching edges count as overlap
    tri = np.array([[3.0, -1.0], [5.0, 1.0], [3.0, 3.0]])
    assert not tpl_amd.intersect_polygons(SQUARE, tri) and tpl_amd.intersect_polygons(SQUARE, tri - [1.5, 0.0])
    assert not tpl_amd.intersect_polygons(np.zeros((0, 2)), SQUARE)               # an empty polygon intersects nothing
    rng = np.random.default_rng(1)
    for _ in range(60):
        a = tpl_amd.convex_hull(rng.uniform(-1, 1, (7, 2)))
        b = tpl_amd.convex_hull(rng.uniform(-1, 1, (6, 2)) + rng.uniform(-2, 2, 2))
        assert tpl_amd.intersect_polygons(a, b) == po.intersect_polygons(a, b)


def test_convex_hull_against_scipy():
    from scipy.spatial import ConvexHull
    got = tpl_amd.convex_hull(np.array([[0, 0], [2, 0], [1, 1], [2, 2], [0, 2], [1, 0.0], [1, 2]]))
    assert np.array_equal(got, SQUARE)                          # colinear edge points dropped, ccw from the lowest-left
    assert np.array_equal(tpl_amd.convex_hull(SQUARE[:3]), SQUARE[:3])            # fewer than four points: returned as is
    rng = np.random.default_rng(2)
    for n in (5, 12, 200):
        pts = rng.normal(size=(n, 2))
        hull = tpl_amd.convex_hull(pts)
        ref = pts[ConvexHull(pts).vertices]                      # ccw, arbitrary start
        start = int(np.argmin(ref[:, 1] * 1e6 + ref[:, 0]))
        assert np.array_equal(hull, np.roll(ref, -start, axis=0))


def test_smooth_path_against_the_restatement():
    s = np.arange(60) * 0.5
    rng = np.random.default_rng(3)
    path = np.stack([s * np.cos(0.3) + rng.normal(0, 0.05, 60), s * np.sin(0.3) + 0.4 * np.sin(s / 4.0) + rng.normal(0, 0.05, 60)], axis=1)
    for kw in (dict(), dict(w_v=0.1, w_a=5.0, w_j=20.0), dict(closed=True)):
        got = tpl_amd.smooth_path(path, 0.5, **kw)
        ref = po.smooth_path(path, 0.5, **kw)
        assert got.shape == (60, 2) and np.allclose(got, ref, rtol=1e-9, atol=1e-9)
    sm = tpl_amd.smooth_path(path, 0.5, w_j=50.0)
    assert np.allclose(sm[0], path[0]) and np.allclose(sm[-1], path[-1], atol=1e-3)          # both ends are pinned
    rough = lambda p: np.abs(np.diff(p, 3, axis=0)).sum()
    assert rough(sm) < 0.2 * rough(path)                                                   # and the jerk is taken out
    with pytest.raises(ValueError):
        tpl_amd.smooth_path(path[:1], 0.5)
