"""DP (s, ds, l) planner, CPU side: hand-derivable cases of the oracle's host-side pieces and the
product's host entry points (which need no GPU) against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lat_lon as oll
import tpl_amd
from tpl_amd._capi import TplLatLonCartState, TplLatLonState


def _state(**kw):
    s = TplLatLonState()
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def test_dynamics_closed_form_and_clamps():
    x0 = _state(t=1.0, s=10.0, ds=4.0, l=0.5)
    out = TplLatLonState()
    oll.lib().llo_dynamics(C.byref(x0), 1.0, -0.5, 2.0, C.byref(out))
    assert (out.t, out.s, out.ds, out.dds, out.l, out.dl) == (3.0, 20.0, 6.0, 1.0, -0.5, -0.5)
    oll.lib().llo_dynamics(C.byref(x0), -3.0, 0.0, 2.0, C.byref(out))          # would move backwards
    assert out.ds == 0.0 and out.s == 12.0
    oll.lib().llo_dynamics(C.byref(x0), -9.0, 0.0, 2.0, C.byref(out))
    assert out.s == 10.0                                                       # never behind x0.s
    prod = TplLatLonState()
    for dds, dl, dt in [(1.0, -0.5, 2.0), (-3.0, 0.0, 2.0), (0.37, 1.9, 0.1)]:
        oll.lib().llo_dynamics(C.byref(x0), dds, dl, dt, C.byref(out))
        tpl_amd._capi.load().tplhip_lat_lon_dynamics(C.byref(x0), dds, dl, dt, C.byref(prod))
        assert bytes(out) == bytes(prod)


def _traj(n=6):
    arr = (TplLatLonState * n)()
    rng = np.random.default_rng(1)
    s = 0.0
    for i in range(n):
        arr[i].t = 0.0 if i == 0 else 1.0 + (i - 1)
        arr[i].s, arr[i].ds, arr[i].dds = s, 5.0 + i, rng.uniform(-1, 1)
        arr[i].l, arr[i].dl = rng.uniform(-1, 1), rng.uniform(-0.5, 0.5)
        arr[i].cost, arr[i].constr = rng.uniform(0, 10), rng.uniform(0, 1)
        s += 6.0 + i
    return arr


def test_traj_state_and_lerp_match_oracle_bit_for_bit():
    arr = _traj()
    a, b = TplLatLonState(), TplLatLonState()
    lib = tpl_amd._capi.load()
    for t in np.linspace(-0.5, 6.0, 40):
        lib.tplhip_lat_lon_traj_state(arr, len(arr), float(t), C.byref(a))
        oll.lib().llo_traj_state(arr, len(arr), float(t), C.byref(b))
        assert bytes(a) == bytes(b)
        lib.tplhip_lat_lon_traj_lerp(arr, len(arr), float(t), C.byref(a))
        oll.lib().llo_traj_lerp(arr, len(arr), float(t), C.byref(b))
        assert bytes(a) == bytes(b)
    lib.tplhip_lat_lon_traj_lerp(arr, len(arr), 1.5, C.byref(a))
    assert a.s == pytest.approx(0.5 * (arr[1].s + arr[2].s))


def test_cart_lerp_matches_oracle():
    n = 5
    arr = (TplLatLonCartState * n)()
    for i in range(n):
        arr[i].t, arr[i].x, arr[i].y = i * 0.5, 3.0 * i, 0.1 * i * i
        arr[i].heading = 3.0 + 0.1 * i if i < 3 else -3.1 + 0.1 * i            # crosses +-pi
        arr[i].v, arr[i].distance = 5.0 + i, 3.0 * i
    a, b = TplLatLonCartState(), TplLatLonCartState()
    for t in np.linspace(-0.2, 2.4, 30):
        tpl_amd._capi.load().tplhip_lat_lon_cart_lerp(arr, n, float(t), C.byref(a))
        oll.lib().llo_cart_lerp(arr, n, float(t), C.byref(b))
        assert bytes(a) == bytes(b)


def test_smoother_tracks_a_constant_speed_reference():
    """x_ref = (s0 + v t, v, 0, 0), started on the reference: the LQR pass stays on it."""
    p = tpl_amd.DynProgLatLonPlannerParams()
    arr = (TplLatLonState * 4)()
    for i in range(4):
        arr[i].t, arr[i].s, arr[i].ds, arr[i].l = i * 1.0, 8.0 * i, 8.0, 0.7
    x0 = _state(s=0.0, ds=8.0, l=0.7)
    out = (TplLatLonState * 64)()
    n = oll.lib().llo_update_traj_smooth(C.byref(p._c), arr, 4, C.byref(x0), out, 64)
    assert n == 30
    got = oll.states_np(out[:n])
    assert np.allclose(got[:, 1], 0.8 * np.arange(n), atol=1e-4) and np.allclose(got[:, 2], 8.0, atol=1e-4)
    assert np.allclose(got[:, 5], 0.7, atol=1e-5)

