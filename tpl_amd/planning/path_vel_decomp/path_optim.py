"""``PathOptim`` (library/tpl/planning/path_vel_decomp/path_optim.py): the lateral path optimisation around
``lateral_profile`` -- corridor from the lane bounds, evasion of predicted objects, ramped corridor references,
one ``Optim.update()``, Frenet -> Cartesian."""
import ctypes as C

import numpy as np

from ... import _capi
from .. import geometry as geo
from ._common import make_solver, runtime

_dp = C.POINTER(C.c_double)


def rampify_profile(step, horizon, evasion_sharpness, proj_distance, path, gap, lower, upper):
    """path_optim.py:11-59 (numba-compiled there, a host routine of libtplhip.so here): a lower corridor bound that
    rises towards obstacles with a speed-dependent slope; the maximum of a forward and a backward pass."""
    path = np.ascontiguousarray(path, dtype=np.float64)
    lower = np.ascontiguousarray(lower, dtype=np.float64)
    upper = np.ascontiguousarray(upper, dtype=np.float64)
    out = np.zeros(len(path))
    col = path[:, 5:]
    _capi.check(_capi.load().tplhip_rampify_lateral_profile(
        float(step), int(horizon), float(evasion_sharpness), float(proj_distance),
        col.ctypes.data_as(_dp), path.shape[1], len(path), float(gap), lower.ctypes.data_as(_dp),
        upper.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
    return out


class CostFunctionParams:
    """path_optim.py:62-69"""

    def __init__(self):
        self.w_d = 0.5
        self.w_v_d = 0.5
        self.w_a_d = 0.5
        self.w_k = 0.5


class Params:
    """path_optim.py:72-97"""

    def __init__(self):
        self.horizon = 250
        self.step = 0.5
        self.ref_step = 0.5
        self.min_d_dd = -2.5
        self.max_d_dd = 2.5
        self.max_lat_acc = 2.5
        self.lateral_min_gap = 2.0
        self.offset_center_line = 0.0
        self.fix_seconds = 3.0
        self.evasion_lon_d_safe = 4
        self.evasion_lat_d_safe = 2.0
        self.evasion_lat_d_comf = 2.0
        self.evasion_sharpness = 20.0
        self.cost_func = CostFunctionParams()


class PathOptim:

    def __init__(self, solver=None):
        self.opt = make_solver("lateral_profile", solver)                        # :102-106
        self.opt.lg_mult_limit = 0.0
        self.opt.barrier_weight[:] = 1000.0
        self.opt.lagrange_multiplier[:] = 0.0
        self.opt_path = np.zeros((1, 6))
        self.d_lower_constr = np.zeros((0, 1))
        self.d_upper_constr = np.zeros((0, 1))
        self.reset_counter = 0
        self.reset_required = False
        self.prev_time = -10000.0

    def _evade(self, env, params, path, d_lower_constr, d_upper_constr, d_mid, l_veh, step):
        """Dynamic objects push the corridor bound of the side they are passed on (:171-250)."""
        local_map, veh = env.local_map, env.vehicle_state
        evade_these = []
        for obj in env.predicted:
            preds = [p for p in obj.predictions if p.uuid_assoc_map == env.local_map_behind.uuid]
            if preds:
                evade_these.append((obj, preds[0]))
        first_stop = np.argmax(path[:, 5] < 0.1)                                 # objects behind a stop point do not count
        if first_stop == 0 and path[0, 5] > 0.1:
            first_stop = 1000
        for obj, pred in evade_these:
            pps = geo.project(path[:, :2], np.asarray(obj.hull, dtype=np.float64).reshape(-1, 2))
            p_min = min(pps, key=lambda p: p.arc_len)
            p_max = max(pps, key=lambda p: p.arc_len)
            if p_min.index > first_stop:
                continue
            v_diff = path[0, 5] - obj.v * pred.cos_angle_dist
            v_scale = path[0, 5] / max(0.001, np.abs(v_diff))
            if v_diff < 0.0:
                v_scale *= -1.0
            lo_arc = (p_min.arc_len - params.evasion_lon_d_safe - l_veh / 2) * v_scale
            hi_arc = (p_max.arc_len + params.evasion_lon_d_safe + l_veh / 2) * v_scale
            i_lo = max(0, min(path.shape[0], int(lo_arc / step)))
            i_hi = max(0, min(path.shape[0], int(hi_arc / step)))
            pps_in = [p for p in pps if p.in_bounds
                      and -local_map.d_right[p.index] <= p.distance <= local_map.d_left[p.index]]
            if not pps_in:
                continue
            evade_side = obj.evade
            if evade_side not in ["right", "left"]:
                if obj.v < path[0, 5] * 0.75 or pred.cos_angle_dist < 0.0:       # slow or oncoming: evade anyway
                    only_left = all(p.distance < d_mid[p.start] for p in pps_in)
                    only_right = all(p.distance > d_mid[p.start] for p in pps_in)
                    if only_left:
                        evade_side = "left"
                    if only_right:
                        evade_side = "right"
            if evade_side == "right":
                d_mult, side, other_side = -1.0, d_upper_constr, d_lower_constr
            elif evade_side == "left":
                d_mult, side, other_side = 1.0, d_lower_constr, d_upper_constr
            else:
                continue
            d_max = max(pps_in, key=lambda p: d_mult * p.distance).distance
            d_max += d_mult * (params.evasion_lat_d_safe + veh.width / 2.0)
            evade_dist = d_mult * d_max
            side[i_lo:i_hi] = np.minimum(-other_side[i_lo:i_hi], np.maximum(side[i_lo:i_hi], evade_dist))

    @runtime
    def update(self, env, params):
        local_map = env.local_map
        path_len = min(params.horizon, local_map.steps_ref)
        path = np.array(local_map.path[:path_len], dtype=np.float64)
        veh = env.vehicle_state
        proj_veh = geo.project(path[:, :2], np.array([veh.x, veh.y]))
        l_veh = veh.rear_axis_to_front + veh.rear_axis_to_rear

        opt = self.opt
        opt.horizon = path_len                                                   # :122-127
        opt.step = params.step
        opt.params.ref_step = local_map.step_size_ref
        opt.u_min[:] = -params.max_d_dd
        opt.u_max[:] = params.max_d_dd
        geo.merge(opt.params, params.cost_func)

        d_lower_constr = -np.asarray(local_map.d_right[:path_len], dtype=np.float64) + veh.width / 2.0
        d_upper_constr = -np.asarray(local_map.d_left[:path_len], dtype=np.float64) + veh.width / 2.0

        # (re-)initialisation via warm starting (:139-170)
        si = local_map.shift_idx_start_ref
        self.reset_required |= self.reset_counter != env.reset_counter
        self.reset_required |= self.prev_time > env.t
        self.reset_counter = env.reset_counter
        self.prev_time = env.t
        length_veh = veh.rear_axis_to_rear + veh.rear_axis_to_front + veh.v * params.fix_seconds
        fix = int(np.ceil(length_veh / local_map.step_size_ref))
        if self.reset_required or not 0 <= si < path_len:
            opt.x[0, 0] = proj_veh.distance                                      # from the reference line
            opt.x[0, 1] = 0.0
            opt.u[:] = 0.0
            opt.params.d_upper_constr = -d_upper_constr
            opt.params.d_lower_constr = d_lower_constr
        else:
            opt.shift(si)
            opt.u_max[:fix] = 0.0                                                # what the vehicle has passed stays fixed
            opt.u_min[:fix] = 0.0
            d_upper_constr[:fix] = -np.asarray(opt.params.d_upper_constr)[si:si + fix]
            d_lower_constr[:fix] = np.asarray(opt.params.d_lower_constr)[si:si + fix]
        self.reset_required = False

        w_path = -d_upper_constr - d_lower_constr
        d_mid = d_lower_constr + w_path / 2.0
        self._evade(env, params, path, d_lower_constr, d_upper_constr, d_mid, l_veh, opt.step)
        self.d_lower_constr = d_lower_constr
        self.d_upper_constr = d_upper_constr

        # ramped corridor references and the target offset (:259-291)
        d_lower_ref = rampify_profile(opt.step, opt.horizon, params.evasion_sharpness, proj_veh.distance, path,
                                      params.lateral_min_gap, d_lower_constr, -d_upper_constr)
        d_upper_ref = -rampify_profile(opt.step, opt.horizon, params.evasion_sharpness, -proj_veh.distance, path,
                                       params.lateral_min_gap, d_upper_constr, -d_lower_constr)
        w_corridor = d_upper_ref - d_lower_ref
        d_trg = d_lower_ref + np.minimum(w_corridor / 2, params.evasion_lat_d_comf)
        center_line_free = (d_lower_ref < params.offset_center_line) & (d_upper_ref > params.offset_center_line)
        d_trg[center_line_free] = params.offset_center_line

        opt.params.k_ref = path[:, 4]                                             # :295-301
        opt.params.d_lower_constr = d_lower_ref
        opt.params.d_upper_constr = d_upper_ref
        opt.params.d_offset = d_trg
        opt.integrator_type = opt.EULER
        opt.update()

        # Frenet -> Cartesian, resampled (:305-309)
        d = np.asarray(opt.x)[:-1, 0]
        path[:, 0] += -np.sin(path[:, 2]) * d
        path[:, 1] += np.cos(path[:, 2]) * d
        path[:, 2] += np.arctan(np.asarray(opt.x)[:-1, 1])
        self.opt_path = geo.resample_path(path, opt.step, opt.horizon)
