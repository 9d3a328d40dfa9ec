"""Python surface of a generated batched iLQR solver library
(``models/libilqr_<problem>.so``, C-ABI ``include/tplhip_ilqr.h``).

:class:`Optim` rebuilds the attribute surface of the CPython type the
reference's template defines (library/tpl/optim/templates/optim.c:1654-1677,
1824-1944): array properties are numpy views into the library's host mirrors
(so ``opt.u_min[:, 0] = -3`` and ``opt.x[0] = x0`` work in place), setters
broadcast (``opt.lg_mult_limit = 0.0``), scalars have the same names and
aliases, ``params`` enumerates the problem's parameters through ``__slots__``.

``batch = 1`` (the default) behaves like the reference object.  ``batch > 1``
adds a leading axis to every array and turns scalars into ``[batch]`` arrays;
all problems are solved by one kernel launch.  There is no CPU solver behind
this class: ``update()`` runs on the GPU or raises.
"""

import ctypes as C
import os

import numpy as np

_dp = C.POINTER(C.c_double)

FIELDS = ["int_step", "x", "u", "next_int_step", "next_x", "next_u", "prev_x", "prev_u", "prev_k",
          "fx", "fu", "lx", "lu", "lxx", "luu", "lux", "g", "k", "K", "lagrange_multiplier",
          "barrier_weight", "lg_mult_limit", "u_max", "u_min"]
SCALARS = ["traj_costs", "iterations", "lg_iterations", "runtime", "alpha", "mu", "mu_step", "min_mu",
           "trajectory_changed", "improved", "termination_condition", "use_quadratic_terms", "dt",
           "max_iterations", "max_lg_iterations", "opt_start", "min_rel_cost_change", "horizon",
           "integrator_type"]
_INT = {"iterations", "lg_iterations", "mu_step", "termination_condition", "max_iterations",
        "max_lg_iterations", "opt_start", "horizon", "integrator_type"}
_BOOL = {"trajectory_changed", "improved", "use_quadratic_terms"}
_ALIASES = {"step": "dt", "T": "horizon", "elapsed_update_time": "runtime"}
H_MAX = 300

# the reference's Optim.__slots__ as Python sees it (optim.c:1736-1782): the format string holds 41 "s" for 42 names,
# so the list ends at max_lg_iterations -- min_rel_cost_change is an attribute but not a slot
# (tests/golden/genopt_surface.json, read off the imported reference)
OPTIM_SLOTS = ["step", "opt_start", "horizon", "int_step", "x", "u", "next_int_step", "next_x", "next_u",
               "prev_x", "prev_u", "prev_k", "fx", "fu", "lx", "lu", "lxx", "luu", "lux", "g", "k", "K",
               "lagrange_multiplier", "barrier_weight", "lg_mult_limit", "u_max", "u_min", "params",
               "traj_costs", "iterations", "lg_iterations", "runtime", "alpha", "mu", "mu_step",
               "trajectory_changed", "improved", "termination_condition", "use_quadratic_terms",
               "max_iterations", "max_lg_iterations"]

_FIELD_ID = {f: i for i, f in enumerate(FIELDS)}
_S_HORIZON = SCALARS.index("horizon")

SIGNATURES = {
    "tplhip_ilqr_model_name": (C.c_char_p, []),
    "tplhip_ilqr_dims": (None, [C.POINTER(C.c_int)] * 4),
    "tplhip_ilqr_param_name": (C.c_char_p, [C.c_int]),
    "tplhip_ilqr_param_is_array": (C.c_int, [C.c_int]),
    "tplhip_ilqr_last_error": (C.c_char_p, []),
    "tplhip_ilqr_fingerprint": (C.c_char_p, []),
    "tplhip_ilqr_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "tplhip_ilqr_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "tplhip_ilqr_destroy": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_batch": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_t_cap": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_ptr": (_dp, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "tplhip_ilqr_scalars": (_dp, [C.c_void_p]),
    "tplhip_ilqr_param_scalar_ptr": (_dp, [C.c_void_p, C.c_int]),
    "tplhip_ilqr_set_param_array": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "tplhip_ilqr_get_param_array": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(_dp), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "tplhip_ilqr_update": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_shift": (C.c_int, [C.c_void_p, C.c_int]),
    "tplhip_ilqr_dynamics": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int, C.c_double, _dp]),
    "tplhip_ilqr_ct_dynamics": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int, C.c_double, _dp]),
    "tplhip_ilqr_upload": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_restore": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_run_resident": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_sync": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_download": (C.c_int, [C.c_void_p]),
    "tplhip_ilqr_timing": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64)]),
}

_libs = {}


def load_model(so_path):
    so_path = os.path.abspath(so_path)
    lib = _libs.get(so_path)
    if lib is None:
        if not os.path.exists(so_path):
            raise ImportError(f"{so_path} is missing: build it with tpl_amd.optim.genopt.build(...) "
                              "or __graft_entry__.build(); there is no CPU fallback")
        lib = C.CDLL(so_path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _libs[so_path] = lib
    return lib


def _check(lib, rc):
    if rc == 0:
        return
    msg = (lib.tplhip_ilqr_last_error() or b"").decode("utf8", "replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -4:
        raise AssertionError(msg)        # the template's check() failures (optim.c:1195-1200)
    raise RuntimeError(msg)


class Params:
    """The generated ``Params`` type (genopt.py:321-417 / optim.c:1397-1476):
    scalar parameters read/write as floats, array parameters are copied in on
    assignment (float64, C order) and read back as views of the solver's copy."""

    def __init__(self, owner):
        object.__setattr__(self, "_o", owner)

    def __getattr__(self, name):
        o = object.__getattribute__(self, "_o")
        i = o._pindex.get(name)
        if i is None:
            raise AttributeError(name)
        if o._pis_array[i]:
            return o._get_param_array(i)
        v = o._pscalar_view(i)
        return float(v[0]) if o._batch == 1 else v

    def __setattr__(self, name, value):
        o = self._o
        i = o._pindex.get(name)
        if i is None:
            raise AttributeError(f"Params has no attribute {name!r}")
        if o._pis_array[i]:
            o._set_param_array(i, value)
        else:
            o._pscalar_view(i)[...] = value

    def __getstate__(self):
        return {n: (np.array(getattr(self, n)) if self._o._pis_array[i] else getattr(self, n))
                for i, n in enumerate(self._o._pnames)}

    def __deepcopy__(self, memo):
        return self.__getstate__()


Params.__slots__ = property(lambda self: list(self._o._pnames))


class Optim:
    EULER, HEUN, RK4 = 0, 1, 2

    def __init__(self, so_path, batch=1, horizon=None, device=-1, t_cap=None, defaults=None):
        lib = load_model(so_path)
        if t_cap is None:
            # the reference object always has room for H_MAX-1 stages; big batches size
            # their storage by the horizon they are created with
            t_cap = H_MAX - 1 if batch == 1 else (horizon or 20)
        h = C.c_void_p()
        _check(lib, lib.tplhip_ilqr_create(int(device), int(batch), int(t_cap), C.byref(h)))
        d = object.__setattr__
        d(self, "_lib", lib)
        d(self, "_so", so_path)
        d(self, "_h", h)
        d(self, "_batch", int(batch))
        d(self, "_tcap", int(t_cap))
        d(self, "_device", int(device))
        x, u, c, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.tplhip_ilqr_dims(C.byref(x), C.byref(u), C.byref(c), C.byref(n))
        d(self, "_dims", (x.value, u.value, c.value))
        d(self, "_pnames", [lib.tplhip_ilqr_param_name(i).decode() for i in range(n.value)])
        d(self, "_pis_array", [bool(lib.tplhip_ilqr_param_is_array(i)) for i in range(n.value)])
        d(self, "_pindex", {name: i for i, name in enumerate(self._pnames)})
        d(self, "_dyn_buf", None)
        d(self, "_pscalars", {})             # views of the scalar parameters' host mirrors ([batch] doubles, fixed for the solver's life)
        d(self, "_params", Params(self))
        d(self, "_defaults", dict(defaults or {}))
        d(self, "_sc", np.ctypeslib.as_array(lib.tplhip_ilqr_scalars(h), shape=(int(batch), len(SCALARS))))
        if horizon is not None:
            self.horizon = horizon
        for k, v in self._defaults.items():
            setattr(self._params, k, v)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.tplhip_ilqr_destroy(h)
            object.__setattr__(self, "_h", None)

    # ---- arrays -------------------------------------------------------------------
    def _spec(self, f):
        """(rows or None, trailing shape, stored row width)"""
        X, U, Cn = self._dims
        T = int(self._sc[0, SCALARS.index("horizon")])
        CM = max(1, Cn)
        return {
            "int_step": (T + 1, (), 1), "next_int_step": (T + 1, (), 1),
            "x": (T + 1, (X,), X), "next_x": (T + 1, (X,), X), "prev_x": (T + 1, (X,), X),
            "u": (T, (U,), U), "next_u": (T, (U,), U), "prev_u": (T, (U,), U), "prev_k": (T, (U,), U),
            "fx": (T, (X, X), X * X), "fu": (T, (X, U), X * U), "lx": (T, (X,), X), "lu": (T, (U,), U),
            "lxx": (T, (X, X), X * X), "luu": (T, (U, U), U * U), "lux": (T, (U, X), U * X),
            "g": (T, (U,), U), "k": (T, (U,), U), "K": (T, (U, X), U * X),
            "lagrange_multiplier": (T, (Cn,), CM),
            "barrier_weight": (None, (Cn,), CM), "lg_mult_limit": (None, (Cn,), CM),
            "u_max": (T, (U,), U), "u_min": (T, (U,), U),
        }[f]

    def _view(self, f):
        # the call itself matters: it brings the host mirror up to date when the device holds newer values.  The numpy
        # view over the (fixed) mirror is built once per (field, horizon).
        stride = C.c_int64()
        ptr = self._lib.tplhip_ilqr_ptr(self._h, _FIELD_ID[f], C.byref(stride))
        if not ptr:
            _check(self._lib, -2)
        T = int(self._sc[0, _S_HORIZON])
        cache = self.__dict__.setdefault("_views", {})
        hit = cache.get((f, T))
        addr = C.cast(ptr, C.c_void_p).value
        if hit is not None and hit[0] == addr:
            return hit[1]
        v = self._build_view(f, ptr, stride.value)
        cache[(f, T)] = (addr, v)
        return v

    def _build_view(self, f, ptr, stride_value):
        rows, trail, width = self._spec(f)
        stride = C.c_int64(stride_value)
        B = self._batch
        cap_rows = stride.value // width
        shape = (B,) + ((rows,) if rows is not None else ()) + tuple(trail)
        if 0 in shape:                                         # C == 0: nothing stored
            v = np.zeros(shape, dtype=np.float64)
        else:
            full = np.ctypeslib.as_array(ptr, shape=(B, cap_rows, width))
            v = full[:, :(rows if rows is not None else 1)].reshape(shape)   # a view: only rows are cut
        if B == 1:
            v = v[0]
            return np.squeeze(v) if v.ndim > 1 else v          # optim.c:1319-1324
        sq = tuple(i for i in range(1, v.ndim) if v.shape[i] == 1) if v.ndim > 2 else ()
        return np.squeeze(v, axis=sq) if sq else v

    def _get_param_array(self, i):
        data, ndim, per = _dp(), C.c_int(), C.c_int()
        dims = (C.c_int64 * 8)()
        _check(self._lib, self._lib.tplhip_ilqr_get_param_array(self._h, i, C.byref(data), C.byref(ndim), dims, C.byref(per)))
        shape = tuple(dims[k] for k in range(ndim.value))
        if ndim.value == 0 or not data:
            return np.zeros((0,), dtype=np.float64)              # dynArrayToPyObj: empty 1-D array
        if per.value:
            shape = (self._batch,) + shape
        return np.ctypeslib.as_array(data, shape=shape)

    def _set_param_array(self, i, value, per_problem=None):
        arr = np.ascontiguousarray(value, dtype=np.float64)
        if per_problem is None:
            per_problem = False
        dims_src = arr.shape[1:] if per_problem else arr.shape
        if per_problem and arr.shape[0] != self._batch:
            raise ValueError(f"per-problem parameter needs a leading axis of {self._batch}, got {arr.shape}")
        dims = (C.c_int64 * 8)(*dims_src)
        _check(self._lib, self._lib.tplhip_ilqr_set_param_array(
            self._h, i, arr.ctypes.data_as(_dp), len(dims_src), dims, 1 if per_problem else 0))

    def set_param_per_problem(self, name, value):
        """batch extension: array parameter with one array per problem, value[batch, ...]."""
        self._set_param_array(self._pnames.index(name), value, per_problem=True)

    # ---- attribute surface ----------------------------------------------------------
    def __getattr__(self, name):
        name = _ALIASES.get(name, name)
        if name in FIELDS:
            return self._view(name)
        if name in SCALARS:
            col = self._sc[:, SCALARS.index(name)]
            if self._batch > 1:
                return col
            v = col[0]
            return int(v) if name in _INT else (bool(v) if name in _BOOL else float(v))
        if name == "params":
            return self._params
        raise AttributeError(name)

    def __setattr__(self, name, value):
        name = _ALIASES.get(name, name)
        if name in FIELDS:
            view = self._view(name)
            val = np.asarray(value, dtype=np.float64)
            if self._batch == 1 and val.ndim > 1:
                val = np.squeeze(val)                          # optim.c:1334-1336
            view[...] = val
        elif name == "horizon":
            t = np.asarray(value).astype(np.int64)
            self._sc[:, SCALARS.index("horizon")] = np.clip(t, 1, min(H_MAX - 1, self._tcap))   # optim.c:1726-1734
        elif name in SCALARS:
            self._sc[:, SCALARS.index(name)] = value
        elif name == "params":
            # set_params optim.c:1689-1693: copy every parameter of another Params object / state dict
            state = value if isinstance(value, dict) else value.__getstate__()
            for k, v in state.items():
                setattr(self._params, k, v)
        else:
            raise AttributeError(f"Optim has no attribute {name!r}")

    @property
    def batch(self):
        return self._batch

    # ---- methods ------------------------------------------------------------------------
    def update(self):
        _check(self._lib, self._lib.tplhip_ilqr_update(self._h))

    def shift(self, amount):
        _check(self._lib, self._lib.tplhip_ilqr_shift(self._h, int(amount)))

    def _pscalar_view(self, i):
        v = self._pscalars.get(i)
        if v is None:
            v = np.ctypeslib.as_array(self._lib.tplhip_ilqr_param_scalar_ptr(self._h, i), shape=(self._batch,))
            self._pscalars[i] = v
        return v

    def _dyn(self, fn, x, u, t, dt, problem=0):
        X, U, _ = self._dims
        buf = self._dyn_buf
        if buf is None:                     # argument buffers with their pointers made once (a controller calls this ten times a tick)
            bx, bu, bo = np.empty(X), np.empty(U), np.empty(X)
            buf = (bx, bu, bo, bx.ctypes.data_as(_dp), bu.ctypes.data_as(_dp), bo.ctypes.data_as(_dp))
            object.__setattr__(self, "_dyn_buf", buf)
        shape_x, shape_u = np.shape(x), np.shape(u)
        if shape_x != (X,):
            raise ValueError(f'Expected "x_arr" with shape ({X}), but found {shape_x}')
        if shape_u != (U,):
            raise ValueError(f'Expected "u_arr" with shape ({U}), but found {shape_u}')
        buf[0][:] = x
        buf[1][:] = u
        _check(self._lib, fn(self._h, int(problem), buf[3], buf[4], int(t), float(dt), buf[5]))
        return buf[2].copy()

    def dynamics(self, x, u, t, dt, problem=0):
        return self._dyn(self._lib.tplhip_ilqr_dynamics, x, u, t, dt, problem)

    def ct_dynamics(self, x, u, t, dt, problem=0):
        return self._dyn(self._lib.tplhip_ilqr_ct_dynamics, x, u, t, dt, problem)

    def __getstate__(self):
        # pyOptimGetState optim.c:1799-1820
        return {
            "u_min": np.array(self.u_min), "u_max": np.array(self.u_max), "horizon": self.horizon,
            "opt_start": self.opt_start, "barrier_weight": np.array(self.barrier_weight),
            "lg_mult_limit": np.array(self.lg_mult_limit), "max_iterations": self.max_iterations,
            "max_lg_iterations": self.max_lg_iterations, "step": self.dt,
            "use_quadratic_terms": self.use_quadratic_terms, "params": self._params.__getstate__(),
        }

    def __deepcopy__(self, memo):
        # pyOptimDeepCopy optim.c:1784-1797: a new solver object with the same state
        other = Optim(self._so, batch=self._batch, device=self._device, t_cap=self._tcap)
        other._sc[...] = self._sc
        for f in FIELDS:
            getattr(other, f)[...] = getattr(self, f)
        for i, n in enumerate(self._pnames):
            if self._pis_array[i]:
                a = self._get_param_array(i)
                if a.size:
                    other._set_param_array(i, a)
            else:
                setattr(other._params, n, getattr(self._params, n))
        return other

    # ---- device-resident stepping (bench.py) -----------------------------------------------
    def upload(self):
        _check(self._lib, self._lib.tplhip_ilqr_upload(self._h))

    def restore(self):
        _check(self._lib, self._lib.tplhip_ilqr_restore(self._h))

    def run_resident(self):
        _check(self._lib, self._lib.tplhip_ilqr_run_resident(self._h))

    def sync(self):
        _check(self._lib, self._lib.tplhip_ilqr_sync(self._h))

    def download(self):
        _check(self._lib, self._lib.tplhip_ilqr_download(self._h))

    def info(self):
        import json
        buf = C.create_string_buffer(512)
        _check(self._lib, self._lib.tplhip_ilqr_info(self._h, buf, 512))
        return json.loads(buf.value.decode())

    def timing(self, reset=False):
        ms, n = C.c_double(), C.c_int64()
        _check(self._lib, self._lib.tplhip_ilqr_timing(self._h, 1 if reset else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value


Optim.__slots__ = property(lambda self: list(OPTIM_SLOTS))


def factory(so_path, defaults=None):
    """Callable creating solver objects for one compiled problem (what
    ``genopt.build`` returns; the reference's ``get_opt_builder`` genopt.py:622-647)."""
    def make(batch=1, horizon=None, device=-1, t_cap=None):
        return Optim(so_path, batch=batch, horizon=horizon, device=device, t_cap=t_cap, defaults=defaults)
    make.so_path = so_path
    return make
