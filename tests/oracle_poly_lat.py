"""ctypes wrapper around oracle/libpoly_lat_oracle.so (+ oracle/_ref/libpoly_interp_ref.so, the
reference's own quintic built from its source).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from tpl_amd._capi import TplDpEnvParams, TplPolyLatParams, TplPolyLatTrajPoint, TplPolyQuintic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libpoly_lat_oracle.so")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libpoly_interp_ref.so")

_lib = None
_ref = None
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libpoly_lat_oracle.so"])
        l = C.CDLL(LIB)
        PP, EP, TP, QP = (C.POINTER(TplPolyLatParams), C.POINTER(TplDpEnvParams), C.POINTER(TplPolyLatTrajPoint),
                          C.POINTER(TplPolyQuintic))
        l.plo_quintic_init.argtypes = [QP] + [C.c_double] * 8
        l.plo_quintic_eval.restype = C.c_double
        l.plo_quintic_eval.argtypes = [QP, C.c_int, C.c_double]
        l.plo_lattice.argtypes = [PP, EP, TP, _dp, C.c_int, C.c_float, _fp, _fp, _ip]
        l.plo_traj_update.argtypes = [TP, C.c_int]
        l.plo_traj_lerp.argtypes = [TP, C.c_int, C.c_double, TP]
        l.plo_final_path.argtypes = [PP, TP, _dp, C.c_int, C.c_float, _dp, _ip, TP, QP]
        _lib = l
    return _lib


def ref_lib():
    """The reference's PolyQuintic, or None where the reference tree (hence oracle/_ref) is absent."""
    global _ref
    if _ref is None and os.path.exists(REF_LIB):
        l = C.CDLL(REF_LIB)
        l.ref_quintic_init.argtypes = [_dp] + [C.c_double] * 8
        l.ref_quintic_eval.restype = C.c_double
        l.ref_quintic_eval.argtypes = [_dp, C.c_int, C.c_double]
        _ref = l
    return _ref


def cparams(p):
    c = TplPolyLatParams()
    C.memmove(C.byref(c), C.byref(p._c), C.sizeof(c))
    f = np.float32
    if c.s_steps > 1:
        c._s_step_size = (f(c.s_max) - f(c.s_min)) / f(c.s_steps - 1)
    if c.l_dst_steps > 1:
        c._l_dst_step_size = (f(c.l_dst_max) - f(c.l_dst_min)) / f(c.l_dst_steps - 1)
    if c.s_dst_steps > 1:
        c._s_dst_step_size = (f(c.s_dst_max) - f(c.s_dst_min)) / f(c.s_dst_steps - 1)
    return c


def plan(p, ep_c, start, ref_centered, ref_step, offsets, occ):
    """-> (nodes[l_dst, s_dst, s, 4], best (l, s), points as float64[s_steps, 12], quintic)."""
    c = cparams(p)
    nodes = np.zeros((c.l_dst_steps, c.s_dst_steps, c.s_steps, 4), np.float32)
    best = (C.c_int * 2)()
    ref_centered = np.ascontiguousarray(ref_centered, dtype=np.float64)
    occ = np.ascontiguousarray(occ, dtype=np.float32)
    lib().plo_lattice(C.byref(c), C.byref(ep_c), C.byref(start._c), ref_centered.ctypes.data_as(_dp),
                      len(ref_centered), np.float32(ref_step), occ.ctypes.data_as(_fp),
                      nodes.ctypes.data_as(_fp), best)
    pts = (TplPolyLatTrajPoint * c.s_steps)()
    q = TplPolyQuintic()
    off = np.ascontiguousarray(offsets, dtype=np.float64)
    lib().plo_final_path(C.byref(c), C.byref(start._c), ref_centered.ctypes.data_as(_dp), len(ref_centered),
                         np.float32(ref_step), off.ctypes.data_as(_dp), best, pts, C.byref(q))
    return nodes, (best[0], best[1]), points_np(pts), q


FIELDS = ("t", "l", "dl", "ddl", "dddl", "s", "v", "x", "y", "heading", "distance", "k")


def points_np(pts):
    return np.array([[getattr(p, f) for f in FIELDS] for p in pts], dtype=np.float64)
