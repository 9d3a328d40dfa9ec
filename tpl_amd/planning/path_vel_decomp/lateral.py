"""The two lateral stages of tpl's path/velocity-decomposition planner on the MI355X solvers.

* ``PathSmoothing`` -- the reference line through ``ref_line_smoother_k`` (behaviour of
  planning/path_vel_decomp/path_smoothing.py:22-83): used when obstacle avoidance is switched off.
* ``PathOptim``     -- the lateral offset profile through ``lateral_profile`` (path_optim.py:100-307): a corridor from
  the lane bounds, narrowed where predicted objects are passed, softened into references the solver can follow, one
  solve, and the offsets applied to the reference line.

Both warm-start by shifting the previous solution along with the map window and re-initialise from the reference line
when the environment says so."""
import ctypes as C

import numpy as np

from ... import _capi
from .. import geometry as geo
from .._kit import ParamBlock
from ._common import make_solver, runtime

_dp = C.POINTER(C.c_double)
PATH_V_MAX, PATH_K, PATH_HEADING = 5, 4, 2          # columns of a local-map path row (x, y, heading, s, k, v_max)


def rampify_profile(step, horizon, evasion_sharpness, proj_distance, path, gap, lower, upper):
    """A lower corridor bound that approaches obstacles with a slope the vehicle can follow at the path's speed:
    the maximum of a forward and a backward sweep over the stations (the reference compiles this with numba,
    path_optim.py:11-59; here it is a host routine of libtplhip.so)."""
    path, lower, upper = (np.ascontiguousarray(a, dtype=np.float64) for a in (path, lower, upper))
    out = np.zeros(len(path))
    speeds = path[:, PATH_V_MAX:]
    _capi.check(_capi.load().tplhip_rampify_lateral_profile(
        float(step), int(horizon), float(evasion_sharpness), float(proj_distance), speeds.ctypes.data_as(_dp),
        path.shape[1], len(path), float(gap), lower.ctypes.data_as(_dp), upper.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
    return out


class _WarmStart:
    """When may the previous solution be reused?  Not after a reset request, and only if the map window moved by a
    sensible number of reference steps."""

    def __init__(self):
        self.reset_counter = 0
        self.reset_required = False

    def usable(self, env, shift, length, also_reset=False):
        stale = self.reset_required or also_reset or self.reset_counter != env.reset_counter
        self.reset_counter = env.reset_counter
        self.reset_required = False
        return not stale and 0 <= shift < length


# =================================================================================================================
class SmoothingParams(ParamBlock):
    DEFAULTS = dict(horizon=250, step=0.5, ref_step=0.5, k_min=-1.0, k_max=1.0, w_pos=1.0, w_k=0.1)


class PathSmoothing:

    def __init__(self, solver=None):
        opt = self.opt = make_solver("ref_line_smoother_k", solver)
        opt.lg_mult_limit = 0.1
        opt.barrier_weight[:] = 1000.0
        opt.lagrange_multiplier[:] = 0.0
        self.opt_path = np.zeros((1, 6))
        self._warm = _WarmStart()

    reset_counter = property(lambda self: self._warm.reset_counter)
    reset_required = property(lambda self: self._warm.reset_required,
                              lambda self, value: setattr(self._warm, "reset_required", bool(value)))

    @runtime
    def update(self, env, params):
        lm = env.local_map
        n = min(params.horizon, lm.steps_ref)
        line = lm.path[:n]
        opt = self.opt
        opt.horizon, opt.step, opt.integrator_type = n, params.step, opt.EULER
        opt.u_min[:], opt.u_max[:] = params.k_min, params.k_max            # the input is the curvature
        q = opt.params
        q.w_pos, q.w_k, q.ref_step = params.w_pos, params.w_k, lm.step_size_ref
        q.ref_x, q.ref_y = line[:, 0], line[:, 1]
        if self._warm.usable(env, lm.shift_idx_start_ref, n):
            opt.shift(lm.shift_idx_start_ref)
        else:                                                               # start on the reference line itself
            opt.x[0, :] = line[0, :3]
            opt.u[:] = line[:n, PATH_K]
        opt.update()
        smooth = np.zeros((n, 6))
        smooth[:, :3] = opt.x[:-1, :3]
        smooth[:, 3], smooth[:, PATH_K], smooth[:, PATH_V_MAX] = line[:, 3], opt.u, line[:, PATH_V_MAX]
        self.opt_path = smooth


# =================================================================================================================
class LateralCostParams(ParamBlock):
    DEFAULTS = dict(w_d=0.5, w_v_d=0.5, w_a_d=0.5, w_k=0.5)


class OptimParams(ParamBlock):
    DEFAULTS = dict(horizon=250, step=0.5, ref_step=0.5, min_d_dd=-2.5, max_d_dd=2.5, max_lat_acc=2.5, lateral_min_gap=2.0,
                    offset_center_line=0.0, fix_seconds=3.0, evasion_lon_d_safe=4, evasion_lat_d_safe=2.0,
                    evasion_lat_d_comf=2.0, evasion_sharpness=20.0, cost_func=LateralCostParams)


class _Corridor:
    """Lateral bounds of the vehicle's reference point per station, in the solver's sign convention: ``lower`` is the
    right bound, ``upper`` is MINUS the left bound (so that both are "lower bounds" of d and -d)."""

    def __init__(self, lm, n, half_width):
        self.lower = -np.asarray(lm.d_right[:n], dtype=np.float64) + half_width
        self.upper = -np.asarray(lm.d_left[:n], dtype=np.float64) + half_width

    @property
    def middle(self):
        return self.lower + (-self.upper - self.lower) / 2.0

    def push(self, side, lo, hi, offset):
        """Move one bound to ``offset`` over stations lo..hi, never past the opposite bound."""
        mine, other = (self.lower, self.upper) if side == "left" else (self.upper, self.lower)
        mine[lo:hi] = np.minimum(-other[lo:hi], np.maximum(mine[lo:hi], offset))


class PathOptim:

    def __init__(self, solver=None):
        opt = self.opt = make_solver("lateral_profile", solver)
        opt.lg_mult_limit = 0.0
        opt.barrier_weight[:] = 1000.0
        opt.lagrange_multiplier[:] = 0.0
        self.opt_path = np.zeros((1, 6))
        self.d_lower_constr = np.zeros((0, 1))
        self.d_upper_constr = np.zeros((0, 1))
        self.prev_time = -10000.0
        self._warm = _WarmStart()

    reset_counter = property(lambda self: self._warm.reset_counter)
    reset_required = property(lambda self: self._warm.reset_required,
                              lambda self, value: setattr(self._warm, "reset_required", bool(value)))

    # ---- obstacle avoidance -------------------------------------------------------------------------------------
    @staticmethod
    def _passing_side(obj, pred, on_lane, middle, lane_speed):
        """"left" / "right" = the bound that gives way, or None (follow instead of passing).  An explicit request of
        the object wins; otherwise slow or oncoming objects are passed on the side they leave free."""
        if obj.evade in ("right", "left"):
            return obj.evade
        side = None
        if obj.v < lane_speed * 0.75 or pred.cos_angle_dist < 0.0:
            offsets = np.array([p.distance for p in on_lane])
            centre = np.array([middle[p.start] for p in on_lane])
            if np.all(offsets < centre):
                side = "left"
            if np.all(offsets > centre):
                side = "right"
        return side

    def _make_room(self, env, params, path, corridor, vehicle_length, step):
        lm, veh = env.local_map, env.vehicle_state
        lane_speed = path[0, PATH_V_MAX]
        # objects behind the first stop point of the path are somebody else's problem
        stop_at = np.argmax(path[:, PATH_V_MAX] < 0.1)
        if stop_at == 0 and lane_speed > 0.1:
            stop_at = 1000
        middle = corridor.middle
        n = path.shape[0]
        for obj in env.predicted:
            pred = next((p for p in obj.predictions if p.uuid_assoc_map == env.local_map_behind.uuid), None)
            if pred is None:
                continue
            corners = geo.project(path[:, :2], np.asarray(obj.hull, dtype=np.float64).reshape(-1, 2))
            rear = min(corners, key=lambda p: p.arc_len)
            front = max(corners, key=lambda p: p.arc_len)
            if rear.index > stop_at:
                continue
            # the stretch to keep clear, stretched by how long the pass takes at the speed difference
            closing = lane_speed - obj.v * pred.cos_angle_dist
            stretch = lane_speed / max(0.001, np.abs(closing)) * (-1.0 if closing < 0.0 else 1.0)
            gap, half = params.evasion_lon_d_safe, vehicle_length / 2
            lo = int(np.clip(int((rear.arc_len - gap - half) * stretch / step), 0, n))
            hi = int(np.clip(int((front.arc_len + gap + half) * stretch / step), 0, n))
            on_lane = [p for p in corners if p.in_bounds and -lm.d_right[p.index] <= p.distance <= lm.d_left[p.index]]
            if not on_lane:
                continue
            side = self._passing_side(obj, pred, on_lane, middle, lane_speed)
            if side is None:
                continue
            sign = 1.0 if side == "left" else -1.0
            edge = max(on_lane, key=lambda p: sign * p.distance).distance + sign * (params.evasion_lat_d_safe + veh.width / 2.0)
            corridor.push(side, lo, hi, sign * edge)

    # ---- one cycle ------------------------------------------------------------------------------------------------
    @runtime
    def update(self, env, params):
        lm, veh = env.local_map, env.vehicle_state
        n = min(params.horizon, lm.steps_ref)
        path = np.array(lm.path[:n], dtype=np.float64)
        at_vehicle = geo.project(path[:, :2], np.array([veh.x, veh.y]))
        vehicle_length = veh.rear_axis_to_front + veh.rear_axis_to_rear

        opt = self.opt
        opt.horizon, opt.step = n, params.step
        opt.params.ref_step = lm.step_size_ref
        opt.u_min[:], opt.u_max[:] = -params.max_d_dd, params.max_d_dd
        geo.merge(opt.params, params.cost_func)
        corridor = _Corridor(lm, n, veh.width / 2.0)

        shift = lm.shift_idx_start_ref
        went_back = self.prev_time > env.t
        self.prev_time = env.t
        if self._warm.usable(env, shift, n, also_reset=went_back):
            # what the vehicle covers within the next seconds keeps the previous solution: no freedom, old bounds
            frozen = int(np.ceil((vehicle_length + veh.v * params.fix_seconds) / lm.step_size_ref))
            opt.shift(shift)
            opt.u_max[:frozen] = 0.0
            opt.u_min[:frozen] = 0.0
            corridor.upper[:frozen] = -np.asarray(opt.params.d_upper_constr)[shift:shift + frozen]
            corridor.lower[:frozen] = np.asarray(opt.params.d_lower_constr)[shift:shift + frozen]
        else:
            opt.x[0, 0], opt.x[0, 1] = at_vehicle.distance, 0.0
            opt.u[:] = 0.0
            opt.params.d_upper_constr = -corridor.upper
            opt.params.d_lower_constr = corridor.lower

        self._make_room(env, params, path, corridor, vehicle_length, opt.step)
        self.d_lower_constr, self.d_upper_constr = corridor.lower, corridor.upper

        # bounds the solver is asked to respect: the hard corridor softened into ramps, on both sides
        ramp = lambda here, mine, other: rampify_profile(opt.step, opt.horizon, params.evasion_sharpness, here, path,
                                                         params.lateral_min_gap, mine, -other)
        lower_ref = ramp(at_vehicle.distance, corridor.lower, corridor.upper)
        upper_ref = -ramp(-at_vehicle.distance, corridor.upper, corridor.lower)
        # the offset to aim for: a comfortable distance from the right ramp, the lane centre where that is free
        target = lower_ref + np.minimum((upper_ref - lower_ref) / 2, params.evasion_lat_d_comf)
        target[(lower_ref < params.offset_center_line) & (upper_ref > params.offset_center_line)] = params.offset_center_line

        q = opt.params
        q.k_ref, q.d_lower_constr, q.d_upper_constr, q.d_offset = path[:, PATH_K], lower_ref, upper_ref, target
        opt.integrator_type = opt.EULER
        opt.update()

        # apply the offsets: shift every reference point sideways, turn its heading by the offset's slope, resample
        solution = np.asarray(opt.x)[:-1]
        heading = path[:, PATH_HEADING].copy()
        path[:, 0] += -np.sin(heading) * solution[:, 0]
        path[:, 1] += np.cos(heading) * solution[:, 0]
        path[:, PATH_HEADING] += np.arctan(solution[:, 1])
        self.opt_path = geo.resample_path(path, opt.step, opt.horizon)
