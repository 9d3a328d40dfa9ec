"""ctypes view of ``libtplhip.so`` (``include/tplhip.h``).

This module only loads the shared library and declares its signatures.  There
is no Python or CPU fallback for any compute entry point: if the library is
missing, importing this module raises, and if no HIP device is usable the
library itself reports ``TPLHIP_EDEVICE``.
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TPLHIP_LIB: alternative build of the same library (kernel ablations while tuning)
LIB_PATH = os.environ.get("TPLHIP_LIB") or os.path.join(_HERE, "libtplhip.so")

TPLHIP_OK = 0
TPLHIP_EINVAL = -1
TPLHIP_EDEVICE = -2
TPLHIP_ESTATE = -3


class TplDpLonParams(C.Structure):
    """DynProgLonPlannerParams, library/include/tplcpp/dyn_prog/lon_planner.cuh:7-53."""
    _fields_ = [
        ("s_min", C.c_float), ("s_max", C.c_float),
        ("v_min", C.c_float), ("v_max", C.c_float),
        ("a_min", C.c_float), ("a_max", C.c_float),
        ("j_min", C.c_float), ("j_max", C.c_float),
        ("t_steps", C.c_int32), ("s_steps", C.c_int32),
        ("v_steps", C.c_int32), ("a_steps", C.c_int32),
        ("dt_start", C.c_float), ("dt", C.c_float),
        ("_s_step_size", C.c_float), ("_v_step_size", C.c_float), ("_a_step_size", C.c_float),
        ("time_gap", C.c_float), ("gap_min", C.c_float),
        ("w_progress", C.c_float), ("w_a", C.c_float), ("w_j", C.c_float),
        ("w_snap", C.c_float), ("w_safety_dist", C.c_float),
        ("path_step_size", C.c_float), ("path_steps", C.c_int32),
        ("width_veh", C.c_float), ("length_veh", C.c_float),
    ]


class TplLonState(C.Structure):
    """LonState, lon_planner.cuh:55-67."""
    _fields_ = [(n, C.c_float) for n in ("t", "s", "v", "a", "j", "cost", "constr")]


class TplPathState(C.Structure):
    """PathState, library/include/tplcpp/dyn_prog/common.cuh:98-110."""
    _fields_ = [(n, C.c_float) for n in ("x", "y", "s", "l", "k", "v_max", "distance")]


class TplDpEnvParams(C.Structure):
    """DynProgEnvParams, library/include/tplcpp/dyn_prog/env.cuh:11-44."""
    _fields_ = [
        ("t_min", C.c_float), ("t_max", C.c_float),
        ("l_min", C.c_float), ("l_max", C.c_float),
        ("s_min", C.c_float), ("s_max", C.c_float),
        ("ds_max", C.c_float),
        ("dir_min", C.c_float), ("dir_max", C.c_float),
        ("dt", C.c_float), ("dt_start", C.c_float),
        ("t_steps", C.c_int32), ("l_steps", C.c_int32),
        ("s_steps", C.c_int32), ("dir_steps", C.c_int32),
        ("scale_objects", C.c_float), ("dilation", C.c_float),
        ("_l_step_size", C.c_float), ("_s_step_size", C.c_float),
    ]


class TplPolyLatParams(C.Structure):
    """PolyLatPlannerParams, library/include/tplcpp/dyn_prog/poly_lat_planner.cuh:11-59."""
    _fields_ = [
        ("l_min", C.c_float), ("l_max", C.c_float), ("s_min", C.c_float), ("s_max", C.c_float),
        ("s_steps", C.c_int32),
        ("l_dst_min", C.c_float), ("l_dst_max", C.c_float), ("s_dst_min", C.c_float), ("s_dst_max", C.c_float),
        ("l_dst_steps", C.c_int32), ("s_dst_steps", C.c_int32),
        ("l_trg", C.c_float),
        ("w_l", C.c_float), ("w_k", C.c_float), ("w_dl", C.c_float), ("w_ddl", C.c_float),
        ("w_dddl", C.c_float), ("w_right", C.c_float), ("w_len", C.c_float),
        ("k_abs_max", C.c_float), ("a_lat_abs_max", C.c_float),
        ("width_veh", C.c_float), ("length_veh", C.c_float),
        ("_s_step_size", C.c_float), ("_l_dst_step_size", C.c_float), ("_s_dst_step_size", C.c_float),
    ]


class TplPolyLatTrajPoint(C.Structure):
    """PolyLatTrajPoint, poly_lat_planner.cuh:61-76."""
    _fields_ = [
        ("t", C.c_float), ("l", C.c_float), ("dl", C.c_float), ("ddl", C.c_float), ("dddl", C.c_float),
        ("s", C.c_float), ("v", C.c_float),
        ("x", C.c_double), ("y", C.c_double),
        ("heading", C.c_float),
        ("distance", C.c_double),
        ("k", C.c_float),
    ]


class TplPolyQuintic(C.Structure):
    """PolyQuintic, library/include/tplcpp/poly_interp.cuh:34-61."""
    _fields_ = [("x0", C.c_double), ("x1", C.c_double), ("d", C.c_double), ("c", C.c_double * 6)]


assert C.sizeof(TplPolyLatTrajPoint) == 72


class TplProjection(C.Structure):
    """Projection, library/include/tplcpp/utils.hpp:240-256."""
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("alpha", C.c_double), ("index", C.c_int64),
                ("point", C.c_double * 2), ("distance", C.c_double), ("arc_len", C.c_double), ("angle", C.c_double),
                ("tangent", C.c_double * 2), ("in_bounds", C.c_int32)]


class TplLatLonParams(C.Structure):
    """DynProgLatLonPlannerParams, library/include/tplcpp/dyn_prog/lat_lon_planner.cuh:9-62."""
    _fields_ = (
        [(n, C.c_float) for n in ("s_min", "s_max", "ds_min", "ds_max", "l_min", "l_max",
                                  "dds_min", "dds_max", "dl_min", "dl_max")]
        + [(n, C.c_int32) for n in ("t_steps", "s_steps", "ds_steps", "l_steps")]
        + [(n, C.c_float) for n in ("dt", "dt_start", "dt_smooth_traj", "dds_start", "w_dds_start", "angle_start",
                                    "w_angle_start", "l_trg", "w_progress", "w_dds", "w_ddds", "w_l", "w_dl", "w_ddl",
                                    "w_safety_dist", "w_xing_slow", "slope_abs_max", "w_lat_dist", "d_lat_comf",
                                    "time_gap", "gap_min", "t_st_min", "t_st_max", "s_st", "w_spatio_temporal",
                                    "width_veh", "length_veh")])


class TplLatLonState(C.Structure):
    """LatLonState, lat_lon_planner.cuh:81-103."""
    _fields_ = [(n, C.c_float) for n in ("t", "s", "ds", "dds", "ddds", "l", "dl", "ddl", "dddl", "cost", "constr")] \
        + [("flags_constr", C.c_uint8)]


class TplLatLonCartState(C.Structure):
    """LatLonCartState, lat_lon_planner.cuh:64-79."""
    _fields_ = [(n, C.c_double) for n in ("t", "distance", "x", "y", "v", "a", "heading", "k", "cost", "constr")] \
        + [("flags_constr", C.c_uint8)]


assert C.sizeof(TplLatLonState) == 48 and C.sizeof(TplLatLonCartState) == 88

class TplRefLinePoint(C.Structure):
    """RefLinePoint, utils.hpp:129-140."""
    _fields_ = [(n, C.c_double) for n in ("x", "y", "heading", "s", "k", "v_max", "d_left", "d_right", "semantic")]


_P = C.POINTER
_vp = C.c_void_p
_fp = _P(C.c_float)
_dp = _P(C.c_double)

# name -> (restype, argtypes); mirrors include/tplhip.h declaration by declaration
SIGNATURES = {
    "tplhip_last_error": (C.c_char_p, []),
    "tplhip_device_count": (C.c_int, []),
    "tplhip_set_device": (C.c_int, [C.c_int]),
    "tplhip_ref_line_center": (C.c_int, [_dp, C.c_int, C.c_int, _dp, _dp, _dp]),
    "tplhip_gen_prediction_geometry": (C.c_int, [_dp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_double,
                                                 C.c_double, C.c_double, _dp, C.c_int, _P(C.c_int)]),
    "tplhip_point_in_polygon": (C.c_int, [_dp, C.c_int, _dp, C.c_int, _P(C.c_uint8)]),
    "tplhip_intersect_polygons": (C.c_int, [_dp, C.c_int, _dp, C.c_int, _P(C.c_int)]),
    "tplhip_convex_hull": (C.c_int, [_dp, C.c_int, _dp, _P(C.c_int)]),
    "tplhip_smooth_path": (C.c_int, [_dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _dp]),
    "tplhip_lerp": (C.c_int, [_dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp]),
    "tplhip_ref_line_lerp": (C.c_int, [_dp, C.c_int, C.c_float, C.c_float, _dp]),
    "tplhip_version": (C.c_char_p, []),
    "tplhip_dp_lon_params_default": (None, [_P(TplDpLonParams)]),
    "tplhip_dp_env_params_default": (None, [_P(TplDpEnvParams)]),
    "tplhip_lon_dynamics": (None, [_P(TplLonState), C.c_float, C.c_float, _P(TplLonState)]),
    "tplhip_lon_traj_state": (C.c_int, [_P(TplLonState), C.c_int, C.c_float, _P(TplLonState)]),
    "tplhip_lon_traj_state_rows": (C.c_int, [_P(TplLonState), C.c_int, _dp, C.c_int, _dp]),
    "tplhip_dp_env_create": (C.c_int, [C.c_int, _P(_vp)]),
    "tplhip_dp_env_destroy": (C.c_int, [_vp]),
    "tplhip_dp_env_reinit": (C.c_int, [_vp, _P(TplDpEnvParams)]),
    "tplhip_dp_env_set_occ_map": (C.c_int, [_vp, _fp]),
    "tplhip_dp_env_update_dist_maps": (C.c_int, [_vp]),
    "tplhip_dp_env_get_occ_map": (C.c_int, [_vp, _fp]),
    "tplhip_dp_env_get_dist_map_path": (C.c_int, [_vp, _fp]),
    "tplhip_dp_env_set_ref_line": (C.c_int, [_vp, _P(C.c_double), C.c_int, C.c_float]),
    "tplhip_dp_env_ref_line_offset": (C.c_int, [_vp, _P(C.c_double), _P(C.c_double)]),
    "tplhip_dp_env_insert_geometry": (C.c_int, [_vp, _P(C.c_double), C.c_int, C.c_int]),
    "tplhip_dp_env_update": (C.c_int, [_vp]),
    "tplhip_dp_env_cart_size": (C.c_int, [_vp, _P(C.c_int)]),
    "tplhip_dp_env_get_occ_map_cartesian": (C.c_int, [_vp, _P(C.c_uint32)]),
    "tplhip_dp_env_get_dist_map_lon": (C.c_int, [_vp, _fp]),
    "tplhip_dp_env_get_occ_map_slices": (C.c_int, [_vp, _fp]),
    "tplhip_dp_env_get_dist_map_lon_slices": (C.c_int, [_vp, _fp]),
    "tplhip_dp_env_get_dist": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, _fp]),
    "tplhip_lat_lon_params_default": (None, [_P(TplLatLonParams)]),
    "tplhip_lat_lon_dynamics": (None, [_P(TplLatLonState), C.c_float, C.c_float, C.c_float, _P(TplLatLonState)]),
    "tplhip_lat_lon_traj_state": (C.c_int, [_P(TplLatLonState), C.c_int, C.c_float, _P(TplLatLonState)]),
    "tplhip_lat_lon_traj_lerp": (C.c_int, [_P(TplLatLonState), C.c_int, C.c_float, _P(TplLatLonState)]),
    "tplhip_lat_lon_cart_lerp": (C.c_int, [_P(TplLatLonCartState), C.c_int, C.c_float, _P(TplLatLonCartState)]),
    "tplhip_lat_lon_create": (C.c_int, [C.c_int, _P(_vp)]),
    "tplhip_lat_lon_destroy": (C.c_int, [_vp]),
    "tplhip_lat_lon_reinit": (C.c_int, [_vp, _P(TplLatLonParams), C.c_int]),
    "tplhip_lat_lon_update_traj_dp": (C.c_int, [_vp, _vp, _P(TplLatLonState)]),
    "tplhip_lat_lon_set_filter_rule": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "tplhip_lat_lon_update_traj_smooth": (C.c_int, [_vp, _P(TplLatLonState), C.c_int, _P(TplLatLonState),
                                                    _P(TplLatLonState), C.c_int, _P(C.c_int)]),
    "tplhip_lat_lon_frenet_to_cartesian": (C.c_int, [_vp, _P(TplLatLonState), C.c_int, _P(TplLatLonCartState)]),
    "tplhip_lat_lon_reeval_traj": (C.c_int, [_vp, _vp, _P(TplLatLonState), C.c_int, _P(TplLatLonState)]),
    "tplhip_lat_lon_copy_backward_pass_tex": (C.c_int, [_vp, C.c_int, _fp]),
    "tplhip_lat_lon_query_backward_pass_tex": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float, C.c_float, _fp]),
    "tplhip_poly_lat_params_default": (None, [_P(TplPolyLatParams)]),
    "tplhip_poly_quintic_init": (None, [_P(TplPolyQuintic)] + [C.c_double] * 8),
    "tplhip_poly_quintic_eval": (C.c_double, [_P(TplPolyQuintic), C.c_int, C.c_double]),
    "tplhip_poly_lat_create": (C.c_int, [C.c_int, _P(_vp)]),
    "tplhip_poly_lat_destroy": (C.c_int, [_vp]),
    "tplhip_poly_lat_reinit": (C.c_int, [_vp, _P(TplPolyLatParams)]),
    "tplhip_poly_lat_update": (C.c_int, [_vp, _P(TplPolyLatTrajPoint), _vp, _P(TplPolyLatTrajPoint),
                                         _P(TplPolyQuintic), _P(C.c_int), _P(C.c_int)]),
    "tplhip_poly_lat_copy_nodes": (C.c_int, [_vp, _fp]),
    "tplhip_poly_lat_traj_update": (C.c_int, [_P(TplPolyLatTrajPoint), C.c_int]),
    "tplhip_poly_lat_traj_lerp": (C.c_int, [_P(TplPolyLatTrajPoint), C.c_int, C.c_double, _P(TplPolyLatTrajPoint)]),
    "tplhip_poly_lat_traj_lerp_rows": (C.c_int, [_P(TplPolyLatTrajPoint), C.c_int, _dp, C.c_int, _dp]),
    "tplhip_poly_lat_traj_rows": (C.c_int, [_P(TplPolyLatTrajPoint), C.c_int, _dp]),
    "tplhip_dp_lon_create": (C.c_int, [C.c_int, _P(_vp)]),
    "tplhip_dp_lon_destroy": (C.c_int, [_vp]),
    "tplhip_dp_lon_reinit": (C.c_int, [_vp, _P(TplDpLonParams)]),
    "tplhip_dp_lon_update": (C.c_int, [_vp, _P(TplLonState), _fp, _vp, _P(TplLonState)]),
    "tplhip_dp_lon_update_batch": (C.c_int, [_vp, C.c_int, _P(TplLonState), _fp, _vp, _P(TplLonState)]),
    "tplhip_dp_lon_reeval": (C.c_int, [_vp, _P(TplLonState), C.c_int, _fp, _vp, _P(TplLonState)]),
    "tplhip_dp_lon_copy_table": (C.c_int, [_vp, C.c_int, _fp]),
    "tplhip_dp_lon_upload": (C.c_int, [_vp, C.c_int, _P(TplLonState), _fp]),
    "tplhip_dp_lon_plan_resident": (C.c_int, [_vp, _vp]),
    "tplhip_dp_lon_sync": (C.c_int, [_vp]),
    "tplhip_dp_lon_download": (C.c_int, [_vp, _P(TplLonState)]),
    "tplhip_dp_lon_stage_download": (C.c_int, [_vp, C.c_int]),
    "tplhip_dp_lon_collect_download": (C.c_int, [_vp, C.c_int, _P(TplLonState)]),
    "tplhip_dp_lon_info": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "tplhip_dp_lon_set_filter_rule": (C.c_int, [_vp, C.c_int, C.c_int]),
    "tplhip_dp_lon_set_contraction": (C.c_int, [_vp, C.c_int]),
    "tplhip_dp_lon_sweep_built": (C.c_int, [C.c_int] * 7),
    "tplhip_dp_lon_set_blend_order": (C.c_int, [_vp, C.c_int]),
    "tplhip_dp_lon_set_chained_sweep": (C.c_int, [_vp, C.c_int]),
    "tplhip_dp_lon_timing_enable": (C.c_int, [_vp, C.c_int]),
    "tplhip_dp_lon_timing_get": (C.c_int, [_vp, _P(C.c_double), _P(C.c_int64)]),
    "tplhip_path_project": (C.c_int, [_P(C.c_double), C.c_int, C.c_int, C.c_int, _P(C.c_double), C.c_int, _P(TplProjection)]),
    "tplhip_path_resample": (C.c_int, [_P(C.c_double), C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                       _P(C.c_double), _P(C.c_int)]),
    "tplhip_rampify_velocity_profile": (C.c_int, [C.c_int, C.c_double, C.c_int, C.c_double, _P(C.c_double), C.c_int,
                                                  C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                                  _P(C.c_double)]),
    "tplhip_rampify_lateral_profile": (C.c_int, [C.c_double, C.c_int, C.c_double, C.c_double, _P(C.c_double), C.c_int,
                                                 C.c_int, C.c_double, _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "tplhip_dp_lon_record_bytes": (C.c_size_t, [C.c_int]),
    "tplhip_dp_lon_scenarios_upload": (C.c_int, [_vp, C.c_int, _P(C.c_float), _P(TplDpEnvParams), _P(TplLonState),
                                                 _P(C.c_float), C.c_int, C.c_int]),
    "tplhip_dp_lon_scenarios_set_block": (C.c_int, [_vp, C.c_int, C.c_int, _P(C.c_float), _P(TplLonState)]),
    "tplhip_dp_lon_scenarios_plan": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tplhip_dp_lon_scenarios_download": (C.c_int, [_vp, _P(TplLonState)]),
    "tplhip_dp_lon_scenarios_best": (C.c_int, [_vp, C.c_void_p]),
    "tplhip_dp_lon_scenarios_best_device": (C.c_int, [_vp, _P(C.c_void_p), _P(C.c_size_t)]),
    "tplhip_dp_lon_scenarios_copy_table": (C.c_int, [_vp, C.c_int, C.c_int, _P(C.c_float)]),
}

_lib = None


def load():
    """Load ``libtplhip.so`` (once).  Raises ``ImportError`` if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tpl_amd/csrc`. tpl_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    # (TPLHIP_LIB_PARTIAL=1, tuning only: an OLDER build named by TPLHIP_LIB may lack entry points declared since -- same-box
    # A/Bs against it use what it has; calling a missing one raises AttributeError at the call)
    partial = bool(os.environ.get("TPLHIP_LIB")) and os.environ.get("TPLHIP_LIB_PARTIAL") == "1"
    for name, (res, args) in SIGNATURES.items():
        if partial and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().tplhip_last_error()
    return msg.decode("utf8", "replace") if msg else ""


def check(rc):
    """Map a status code to the exception type the reference raises."""
    if rc == TPLHIP_OK:
        return
    msg = last_error()
    if rc == TPLHIP_EINVAL:
        raise ValueError(msg)
    raise RuntimeError(msg)
