"""ctypes wrapper around oracle/libdp_env_oracle.so (test infrastructure only).

Only tests/ and bench.py's cpu_baseline leg may import this module; the product
(tpl_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from tpl_amd._capi import TplDpEnvParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libdp_env_oracle.so")

_lib = None
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint32)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libdp_env_oracle.so"])
        l = C.CDLL(LIB)
        P = C.POINTER(TplDpEnvParams)
        l.dpo_env_center_ref_line.argtypes = [_dp, C.c_int, _dp, _dp]
        l.dpo_env_insert_geometry.argtypes = [P, _dp, C.c_int, C.c_int, _dp, _fp, _fp, _up]
        l.dpo_env_cart_size.restype = C.c_int
        l.dpo_env_cart_size.argtypes = [P]
        l.dpo_env_render.argtypes = [P, _fp, _fp, _up, C.c_int, _up]
        l.dpo_env_occupancy_to_frenet.argtypes = [P, _dp, C.c_int, C.c_float, _up, _fp]
        l.dpo_env_close_intersections.argtypes = [P, _dp, C.c_int, C.c_float, _fp]
        l.dpo_env_dist_map_lon.argtypes = [P, _fp, _fp]
        l.dpo_env_update.restype = C.c_int
        l.dpo_env_update.argtypes = [P, _dp, C.c_int, C.c_float, _fp, _fp, _up, C.c_int, _up, _fp, _fp]
        _lib = l
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def _u(a):
    return a.ctypes.data_as(_up)


def cparams(p):
    """Copy of the ctypes struct behind a DynProgEnvParams, step sizes filled in (env.cuh:39-43)."""
    c = TplDpEnvParams()
    C.memmove(C.byref(c), C.byref(p._c), C.sizeof(c))
    c._l_step_size = np.float32((np.float32(c.l_max) - np.float32(c.l_min)) / np.float32(c.l_steps - 1))
    c._s_step_size = np.float32((np.float32(c.s_max) - np.float32(c.s_min)) / np.float32(c.s_steps - 1))
    return c


class OracleEnv:
    """The reference's DynProgEnvironment call sequence on the CPU restatement."""

    def __init__(self, params):
        self.c = cparams(params)
        self.ref = np.zeros((0, 9))
        self.step = np.float32(0.0)
        self.offsets = np.zeros(2)
        self.vx, self.vy, self.flags = [], [], []

    def set_ref_line(self, ref_line, step):
        ref_line = np.ascontiguousarray(ref_line, dtype=np.float64)
        self.ref = np.empty_like(ref_line)
        lib().dpo_env_center_ref_line(_d(ref_line), len(ref_line), _d(self.ref), _d(self.offsets))
        self.step = np.float32(step)

    def insert_geometry(self, geom, stationary):
        geom = np.ascontiguousarray(geom, dtype=np.float64)
        n = len(geom)
        vx, vy = np.empty(n, np.float32), np.empty(n, np.float32)
        fl = np.empty(n, np.uint32)
        lib().dpo_env_insert_geometry(C.byref(self.c), _d(geom), n, int(bool(stationary)), _d(self.offsets),
                                      _f(vx), _f(vy), _u(fl))
        self.vx.append(vx); self.vy.append(vy); self.flags.append(fl)

    def update(self):
        """-> (cartesian image, occ[l,s,t], dist_lon[l,s,t,2]); None if the reference would throw."""
        c = self.c
        W = lib().dpo_env_cart_size(C.byref(c))
        vx = np.concatenate(self.vx) if self.vx else np.zeros(0, np.float32)
        vy = np.concatenate(self.vy) if self.vy else np.zeros(0, np.float32)
        fl = np.concatenate(self.flags) if self.flags else np.zeros(0, np.uint32)
        img = np.zeros((W, W), np.uint32)
        occ = np.zeros((c.l_steps, c.s_steps, c.t_steps), np.float32)
        dl = np.zeros((c.l_steps, c.s_steps, c.t_steps, 2), np.float32)
        rc = lib().dpo_env_update(C.byref(c), _d(self.ref), len(self.ref), self.step, _f(vx), _f(vy), _u(fl),
                                  len(vx), _u(img), _f(occ), _f(dl))
        self.vx, self.vy, self.flags = [], [], []
        if rc != 0:
            return None
        return img, occ, dl
