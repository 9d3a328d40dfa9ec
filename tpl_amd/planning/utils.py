"""``tpl.planning.utils`` (library/tpl/planning/utils.py): the velocity-profile ramp."""
import ctypes as C

import numpy as np

from .. import _capi

_dp = C.POINTER(C.c_double)


def rampify_profile(v0, a0, lim_v, a_min, a_max, j_min, j_max, v_min, step):
    """Integrates the spatial velocity dynamics backwards with minimum jerk and acceleration and forwards with
    maximum jerk and acceleration (planning/utils.py:5-65; numba-compiled in the reference, a host routine of
    libtplhip.so here).  ``v0`` / ``a0`` may be None like in the reference.  Returns float64[len(lim_v), 2]."""
    lim_v = np.ascontiguousarray(lim_v, dtype=np.float64)
    out = np.zeros((len(lim_v), 2))
    _capi.check(_capi.load().tplhip_rampify_velocity_profile(
        0 if v0 is None else 1, 0.0 if v0 is None else float(v0), 0 if a0 is None else 1, 0.0 if a0 is None else float(a0),
        lim_v.ctypes.data_as(_dp), len(lim_v), float(a_min), float(a_max), float(j_min), float(j_max), float(v_min),
        float(step), out.ctypes.data_as(_dp)))
    return out
