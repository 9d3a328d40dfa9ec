"""ctypes wrapper around oracle/libilqr_oracle_<problem>.so with the attribute
surface of the reference's Optim type (test infrastructure only; see the header
of oracle/ilqr_oracle.cpp).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FIELDS = ["int_step", "x", "u", "next_int_step", "next_x", "next_u", "prev_x", "prev_u", "prev_k",
          "fx", "fu", "lx", "lu", "lxx", "luu", "lux", "g", "k", "K", "lagrange_multiplier",
          "barrier_weight", "lg_mult_limit", "u_max", "u_min"]
SCALARS = ["traj_costs", "iterations", "lg_iterations", "runtime", "alpha", "mu", "mu_step", "min_mu",
           "trajectory_changed", "improved", "termination_condition", "use_quadratic_terms", "dt",
           "max_iterations", "max_lg_iterations", "opt_start", "min_rel_cost_change", "horizon",
           "integrator_type"]
INT_SCALARS = {"iterations", "lg_iterations", "mu_step", "termination_condition", "max_iterations",
               "max_lg_iterations", "opt_start", "horizon", "integrator_type"}
BOOL_SCALARS = {"trajectory_changed", "improved", "use_quadratic_terms"}
ALIASES = {"step": "dt", "T": "horizon", "elapsed_update_time": "runtime"}

_libs = {}


def load(model, path=None):
    """``path``: another build of the same restatement (bench.py's -O3 -march=native baseline build)."""
    key = (model, path)
    if key in _libs:
        return _libs[key]
    if path is None:
        path = os.path.join(ORACLE_DIR, f"libilqr_oracle_{model}.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, os.path.basename(path)])
    lib = C.CDLL(path)
    lib.ilqo_create.restype = C.c_void_p
    lib.ilqo_create.argtypes = [C.c_int]
    lib.ilqo_destroy.argtypes = [C.c_void_p]
    lib.ilqo_ptr.restype = C.POINTER(C.c_double)
    lib.ilqo_ptr.argtypes = [C.c_void_p, C.c_int]
    lib.ilqo_get.restype = C.c_double
    lib.ilqo_get.argtypes = [C.c_void_p, C.c_int]
    lib.ilqo_set.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.ilqo_param_name.restype = C.c_char_p
    lib.ilqo_param_name.argtypes = [C.c_int]
    lib.ilqo_param_is_array.argtypes = [C.c_int]
    lib.ilqo_set_param_scalar.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.ilqo_set_param_array.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int64)]
    lib.ilqo_update.argtypes = [C.c_void_p]
    lib.ilqo_shift.argtypes = [C.c_void_p, C.c_int]
    dptr = C.POINTER(C.c_double)
    lib.ilqo_dynamics.argtypes = [C.c_void_p, dptr, dptr, C.c_int, C.c_double, dptr]
    lib.ilqo_ct_dynamics.argtypes = [C.c_void_p, dptr, dptr, C.c_int, C.c_double, dptr]
    lib.ilqo_update_many.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
    lib.ilqo_reset_and_update_many.argtypes = [C.POINTER(C.c_void_p), C.c_int, dptr, C.c_int, C.c_int]
    _libs[key] = lib
    return lib


class _Params:
    def __init__(self, owner):
        object.__setattr__(self, "_o", owner)
        object.__setattr__(self, "_vals", {})

    def __setattr__(self, name, value):
        o = self._o
        i = o._pnames.index(name)
        if o._pis_array[i]:
            arr = np.ascontiguousarray(value, dtype=np.float64)
            dims = (C.c_int64 * max(1, arr.ndim))(*arr.shape)
            o._lib.ilqo_set_param_array(o._h, i, arr.ctypes.data_as(C.POINTER(C.c_double)), arr.ndim, dims)
            self._vals[name] = arr.copy()
        else:
            o._lib.ilqo_set_param_scalar(o._h, i, float(value))
            self._vals[name] = float(value)

    def __getattr__(self, name):
        vals = object.__getattribute__(self, "_vals")
        if name in vals:
            return vals[name]
        o = object.__getattribute__(self, "_o")
        if name in o._pnames:
            return np.zeros(0) if o._pis_array[o._pnames.index(name)] else 0.0
        raise AttributeError(name)


_Params.__slots__ = property(lambda self: list(self._o._pnames))


class OracleOptim:
    EULER, HEUN, RK4 = 0, 1, 2

    def __init__(self, model, t_cap=299, lib=None):
        lib = lib or load(model)
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_tcap", t_cap)
        object.__setattr__(self, "_h", lib.ilqo_create(t_cap))
        x, u, c, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.ilqo_dims(C.byref(x), C.byref(u), C.byref(c), C.byref(n))
        object.__setattr__(self, "_dims", (x.value, u.value, c.value))
        object.__setattr__(self, "_pnames", [lib.ilqo_param_name(i).decode() for i in range(n.value)])
        object.__setattr__(self, "_pis_array", [bool(lib.ilqo_param_is_array(i)) for i in range(n.value)])
        object.__setattr__(self, "params", _Params(self))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ilqo_destroy(self._h)
            object.__setattr__(self, "_h", None)

    def _shape(self, f):
        X, U, Cn = self._dims
        T = int(self._lib.ilqo_get(self._h, SCALARS.index("horizon")))
        CM = max(1, Cn)
        return {
            "int_step": ((T + 1,), None), "next_int_step": ((T + 1,), None),
            "x": ((T + 1, X), None), "next_x": ((T + 1, X), None), "prev_x": ((T + 1, X), None),
            "u": ((T, U), None), "next_u": ((T, U), None), "prev_u": ((T, U), None), "prev_k": ((T, U), None),
            "fx": ((T, X, X), None), "fu": ((T, X, U), None), "lx": ((T, X), None), "lu": ((T, U), None),
            "lxx": ((T, X, X), None), "luu": ((T, U, U), None), "lux": ((T, U, X), None),
            "g": ((T, U), None), "k": ((T, U), None), "K": ((T, U, X), None),
            "lagrange_multiplier": ((T, Cn), CM), "barrier_weight": ((Cn,), None), "lg_mult_limit": ((Cn,), None),
            "u_max": ((T, U), None), "u_min": ((T, U), None),
        }[f]

    def _view(self, f):
        shape, row_stride = self._shape(f)
        ptr = self._lib.ilqo_ptr(self._h, FIELDS.index(f))
        if row_stride is not None:                       # lambda rows are stored CMAX wide
            full = np.ctypeslib.as_array(ptr, shape=(shape[0], row_stride))
            v = full[:, :shape[1]]
        else:
            n = int(np.prod(shape)) if len(shape) else 1
            v = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].reshape(shape)
        return np.squeeze(v) if v.ndim > 1 else v         # optim.c:1319-1324 squeezes

    def __getattr__(self, name):
        name = ALIASES.get(name, name)
        if name in FIELDS:
            return self._view(name)
        if name in SCALARS:
            v = self._lib.ilqo_get(self._h, SCALARS.index(name))
            if name in INT_SCALARS:
                return int(v)
            if name in BOOL_SCALARS:
                return bool(v)
            return v
        raise AttributeError(name)

    def __setattr__(self, name, value):
        name = ALIASES.get(name, name)
        if name in FIELDS:
            self._view(name)[...] = np.squeeze(np.asarray(value, dtype=np.float64)) if np.ndim(value) > 1 else value
        elif name in SCALARS:
            self._lib.ilqo_set(self._h, SCALARS.index(name), float(value))
        elif name == "params":
            object.__setattr__(self, name, value)
        else:
            raise AttributeError(name)

    def update(self):
        self._lib.ilqo_update(self._h)

    def shift(self, n):
        self._lib.ilqo_shift(self._h, int(n))

    def _dyn(self, fn, x, u, t, dt):
        X = self._dims[0]
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.empty(X)
        p = C.POINTER(C.c_double)
        fn(self._h, x.ctypes.data_as(p), u.ctypes.data_as(p), int(t), float(dt), out.ctypes.data_as(p))
        return out

    def dynamics(self, x, u, t, dt):
        return self._dyn(self._lib.ilqo_dynamics, x, u, t, dt)

    def ct_dynamics(self, x, u, t, dt):
        return self._dyn(self._lib.ilqo_ct_dynamics, x, u, t, dt)
