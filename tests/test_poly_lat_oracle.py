"""Lateral planner, CPU side: the quintic of oracle/poly_lat_oracle.c and of the product's host
code against the REFERENCE's own PolyQuintic (oracle/_ref, built from library/src/poly_interp.cu
where the reference tree is mounted), and hand-derivable cases of the trajectory helpers."""
import ctypes as C

import numpy as np
import pytest

import oracle_poly_lat as opl
import tpl_amd
from tpl_amd._capi import TplPolyLatTrajPoint, TplPolyQuintic


def _cases(n=200, seed=0):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        x0 = rng.uniform(-50, 300)
        yield (x0, rng.uniform(-5, 5), rng.uniform(-0.5, 0.5), rng.uniform(-0.1, 0.1),
               x0 + rng.uniform(5, 80), rng.uniform(-5, 5), 0.0, 0.0)


def test_quintic_boundary_conditions():
    for args in _cases(20):
        q = tpl_amd.PolyQuintic(*args)
        x0, y0, dy0, ddy0, x1, y1, _, _ = args
        assert q.f(x0) == pytest.approx(y0, abs=1e-9) and q.f(x1) == pytest.approx(y1, abs=1e-9)
        assert q.df(x0) == pytest.approx(dy0, abs=1e-9) and q.df(x1) == pytest.approx(0.0, abs=1e-9)
        assert q.ddf(x0) == pytest.approx(ddy0, abs=1e-9) and q.ddf(x1) == pytest.approx(0.0, abs=1e-9)


def test_quintic_oracle_and_product_match_the_reference_source_bit_for_bit():
    ref = opl.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref/libpoly_interp_ref.so not built (reference tree not mounted)")
    st = (C.c_double * 9)()
    for args in _cases(200, seed=3):
        ref.ref_quintic_init(st, *args)
        qo = TplPolyQuintic()
        opl.lib().plo_quintic_init(C.byref(qo), *args)
        qp = tpl_amd.PolyQuintic(*args)
        want = np.array(list(st))
        assert np.array_equal(want, np.array([qo.x0, qo.x1, qo.d, *qo.c]))
        assert np.array_equal(want, np.array([qp._c.x0, qp._c.x1, qp._c.d, *qp._c.c]))
        for x in np.linspace(args[0] - 1.0, args[4] + 1.0, 7):
            for order in range(4):
                r = ref.ref_quintic_eval(st, order, float(x))
                assert opl.lib().plo_quintic_eval(C.byref(qo), order, float(x)) == r
                assert qp._eval(order, float(x)) == r


def _line_traj(n=11, step=2.0, v=5.0):
    traj = tpl_amd.PolyLatTraj()
    for i in range(n):
        p = tpl_amd.PolyLatTrajPoint()
        p.s, p.x, p.y, p.heading, p.v, p.l = i * step, 3.0 + i * step, -1.0, 0.0, v, 0.1 * i
        traj.points.append(p)
    traj.update_time_dist_curv()
    return traj


def test_update_time_dist_curv_on_a_straight_line():
    traj = _line_traj()
    a = traj.as_numpy()
    assert np.allclose(a[:, 10], np.arange(11) * 2.0)                 # distance
    assert np.allclose(a[:, 0], np.arange(11) * 2.0 / 5.0)            # t = distance / v
    assert np.all(a[:, 11] == 0.0)                                    # curvature of a straight line


def test_lerp_and_insert_after_station():
    traj = _line_traj()
    p = traj.lerp(5.0)
    assert p.s == pytest.approx(5.0) and p.l == pytest.approx(0.25) and p.x == pytest.approx(8.0)
    assert traj.lerp(-3.0).s == pytest.approx(0.0) and traj.lerp(1e3).s == pytest.approx(20.0)      # clipped
    many = traj.lerp(np.array([0.0, 5.0, 20.0]))
    assert many.shape == (3, 12) and many[1, 5] == pytest.approx(5.0)
    other = _line_traj(n=5, step=2.0)
    for q in other.points:
        q.s += 10.0
        q.x += 10.0
    traj.insert_after_station(10.0, other)
    s = traj.as_numpy()[:, 5]
    assert list(s) == [0.0, 2.0, 4.0, 6.0, 8.0, 10.0, 12.0, 14.0, 16.0, 18.0]
    assert np.allclose(traj.as_numpy()[:, 10], np.arange(10) * 2.0)


def test_host_helpers_match_the_oracle_bit_for_bit():
    rng = np.random.default_rng(2)
    n = 40
    pts = (TplPolyLatTrajPoint * n)()
    x = np.cumsum(rng.uniform(0.5, 1.5, n))
    for i in range(n):
        pts[i].x, pts[i].y = x[i], np.sin(0.1 * x[i]) * 3
        pts[i].heading = np.float32(np.arctan(0.3 * np.cos(0.1 * x[i])) + (6.2 if i % 7 == 0 else 0.0))
        pts[i].v, pts[i].s, pts[i].l = rng.uniform(1, 10), x[i], rng.uniform(-1, 1)
    a = (TplPolyLatTrajPoint * n)(*pts)
    b = (TplPolyLatTrajPoint * n)(*pts)
    tpl_amd._capi.load().tplhip_poly_lat_traj_update(a, n)
    opl.lib().plo_traj_update(b, n)
    assert bytes(a) == bytes(b)
    oa, ob = TplPolyLatTrajPoint(), TplPolyLatTrajPoint()
    for d in rng.uniform(-2.0, a[n - 1].distance + 2.0, 50):
        tpl_amd._capi.load().tplhip_poly_lat_traj_lerp(a, n, float(d), C.byref(oa))
        opl.lib().plo_traj_lerp(b, n, float(d), C.byref(ob))
        assert bytes(oa) == bytes(ob)
