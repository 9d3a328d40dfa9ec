"""ctypes wrapper around oracle/libdp_lon_oracle.so (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (tpl_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from tpl_amd._capi import TplDpEnvParams, TplDpLonParams, TplLonState, TplPathState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libdp_lon_oracle.so")

_lib = None
_fp = C.POINTER(C.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libdp_lon_oracle.so"])
        _lib = _declare(C.CDLL(LIB))
    return _lib


def use(path=None):
    """Route this module's helpers through another BUILD of the same restatement: the -O3 -march=native
    build the cpu_baseline leg times (oracle/_native/).  ``None`` returns to the default checker.  (The
    arithmetic rules are run-time state of the one library: see ``rule``.)"""
    global _lib
    _lib = None if path is None else _declare(C.CDLL(path))


# The arithmetic rule (R1 contraction, R2 weight quantisation, R3 blend form and order: header of
# oracle/dp_lon_oracle.c) is a run-time setting of the ONE checker library.  DEFAULT_RULE is what the library
# starts with and what the HIP planner computes by default; names are the ones DESIGN.md / the dump's variant
# table use.
DEFAULT_RULE = dict(quant=1, lerp=0, fma=1, order=0)
NAMED_RULES = {
    "default": {},
    "unfused": dict(fma=0),                       # one rounding per source operation (the default until round 5)
    "wfloor": dict(quant=2), "wnone": dict(quant=0),
    "lerp1": dict(lerp=1), "lerp2": dict(lerp=2),
    "order_sva": dict(order=1), "sum8_f32": dict(order=2), "sum8_wide": dict(order=3),
}


def set_rule(quant=None, lerp=None, fma=None, order=None):
    """Select the rule every later call of the checker follows; arguments left out take DEFAULT_RULE."""
    r = dict(DEFAULT_RULE)
    for k, v in (("quant", quant), ("lerp", lerp), ("fma", fma), ("order", order)):
        if v is not None:
            r[k] = int(v)
    rc = lib().dpo_set_rule(r["quant"], r["lerp"], r["fma"], r["order"])
    assert rc == 0, f"dpo_set_rule refused {r}"
    return r


def get_rule():
    out = (C.c_int * 4)()
    lib().dpo_get_rule(out)
    return dict(quant=out[0], lerp=out[1], fma=out[2], order=out[3])


class rule:
    """``with O.rule(fma=0, order=2): ...`` -- the checker under another rule, the default rule restored on exit.
    ``O.rule("wfloor", fma=0)`` starts from a named rule."""

    def __init__(self, name=None, **kw):
        r = dict(NAMED_RULES[name or "default"])
        r.update(kw)
        self.r = r

    def __enter__(self):
        return set_rule(**self.r)

    def __exit__(self, *exc):
        set_rule()
        return False


def _declare(l):
    if True:
        l.dpo_interp_dist_map_path.restype = C.c_float
        l.dpo_interp_dist_map_path.argtypes = [C.POINTER(TplDpEnvParams), _fp, C.c_float, C.c_float]
        l.dpo_table_floats.restype = C.c_size_t
        l.dpo_lon_dynamics.argtypes = [C.POINTER(TplLonState), C.c_float, C.c_float, C.POINTER(TplLonState)]
        l.dpo_lon_dynamics_host.argtypes = [C.POINTER(TplLonState), C.c_float, C.c_float, C.POINTER(TplLonState)]
        l.dpo_grid_point.restype = C.c_float
        l.dpo_grid_point.argtypes = [C.c_float, C.c_float, C.c_float]
        l.dpo_set_rule.restype = C.c_int
        l.dpo_set_rule.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        l.dpo_get_rule.argtypes = [C.POINTER(C.c_int)]
        l.dpo_interp_path.argtypes = [_fp, C.c_float, C.c_float, C.c_int, C.POINTER(TplPathState)]
        l.dpo_read_node.argtypes = [C.POINTER(TplDpLonParams), _fp, C.c_float, C.c_float, C.c_float, _fp]
        l.dpo_eval_node.argtypes = [C.POINTER(TplDpLonParams), C.POINTER(TplDpEnvParams), C.c_int,
                                    C.POINTER(TplLonState), _fp, _fp, _fp, C.c_float, C.c_int, C.c_int]
        l.dpo_dist_map_path.argtypes = [C.POINTER(TplDpEnvParams), _fp, _fp, C.c_int, C.c_float, _fp]
        l.dpo_backward_stage.argtypes = [C.POINTER(TplDpLonParams), C.POINTER(TplDpEnvParams), C.c_int,
                                         _fp, _fp, _fp, _fp, C.c_int]
        l.dpo_update.restype = C.c_int
        l.dpo_update.argtypes = [C.POINTER(TplDpLonParams), C.POINTER(TplDpEnvParams), _fp, _fp, C.c_int,
                                 C.POINTER(TplLonState), _fp, C.POINTER(TplLonState), C.c_int]
        l.dpo_reeval.restype = C.c_int
        l.dpo_reeval.argtypes = [C.POINTER(TplDpLonParams), C.POINTER(TplDpEnvParams), _fp, _fp, _fp,
                                 C.POINTER(TplLonState), C.c_int, C.POINTER(TplLonState)]
        l.dpo_traj_state.restype = C.c_int
        l.dpo_traj_state.argtypes = [C.POINTER(TplLonState), C.c_int, C.c_float, C.POINTER(TplLonState)]
        l.dpo_update_step_sizes.argtypes = [C.POINTER(TplDpLonParams)]
        l.dpo_env_update_step_sizes.argtypes = [C.POINTER(TplDpEnvParams)]
    return l


def grid_point(x_min, i, step):
    """Grid point i of an axis as the sweep forms it (lon_planner.cu:196-199) under the current rule R1."""
    return lib().dpo_grid_point(np.float32(x_min), np.float32(i), np.float32(step))


def fptr(a):
    return a.ctypes.data_as(_fp)


def cparams(p):
    """Copy of the ctypes struct behind a tpl_amd params object, step sizes filled in."""
    c = type(p._c)()
    C.memmove(C.byref(c), C.byref(p._c), C.sizeof(c))
    if isinstance(c, TplDpLonParams):
        lib().dpo_update_step_sizes(C.byref(c))
    else:
        lib().dpo_env_update_step_sizes(C.byref(c))
    return c


def states_to_np(arr):
    return np.array([[s.t, s.s, s.v, s.a, s.j, s.cost, s.constr] for s in arr], dtype=np.float32)


def dist_map_path(ep, occ, path, path_steps, path_step):
    dist = np.empty((ep.t_steps, ep.s_steps), dtype=np.float32)
    occ = np.ascontiguousarray(occ, dtype=np.float32)
    lib().dpo_dist_map_path(C.byref(ep), fptr(occ), fptr(path), path_steps, path_step, fptr(dist))
    return dist


def update(p, ep, path, occ, inits, nthreads=0, keep_tables=True):
    """Full oracle plan.  p/ep: ctypes structs (cparams).  Returns (tables, trajs, dist)."""
    path = np.ascontiguousarray(path, dtype=np.float32)
    dist = dist_map_path(ep, occ, path, p.path_steps, p.path_step_size)
    tf = lib().dpo_table_floats(C.byref(p))
    tables = np.zeros((p.t_steps, p.a_steps, p.v_steps, p.s_steps, 4), dtype=np.float32)
    assert tables.size == tf * p.t_steps
    n = len(inits)
    cin = (TplLonState * n)()
    for i, s in enumerate(inits):
        C.memmove(C.byref(cin[i]), C.byref(s._c), C.sizeof(TplLonState))
    out = (TplLonState * (n * p.t_steps))()
    rc = lib().dpo_update(C.byref(p), C.byref(ep), fptr(path), fptr(dist), n, cin, fptr(tables), out, nthreads)
    assert rc == 0
    trajs = states_to_np(out).reshape(n, p.t_steps, 7)
    return tables, trajs, dist
