"""``VelocityOptim``: the speed profile over arc length on the MI355X solver (problem ``velocity_profile_space``;
behaviour of planning/path_vel_decomp/velocity_optim.py:58-300).

Per cycle: the previous solution is re-interpolated onto the stations of the shifted path; the closest vehicle on the
path and the manoeuvre module's constraints are cut into the speed limit; a reference the vehicle can follow is ramped
under that limit; crossing-time windows become per-station time bounds; one solve; stations behind the first stop are
masked out."""
import numpy as np
from scipy.interpolate import interp1d

from .. import geometry as geo
from .._kit import ParamBlock
from ..utils import rampify_profile
from ._common import make_solver, runtime

NO_LEADER = 10.0**6                 # station of "nobody ahead"
UNBOUNDED_TIME = 10e10


class TimeConstr:
    """A position that must not be passed before / after time ``t``, with its projection on the current path."""

    def __init__(self, t=0.0, pos=None):
        self.pos = np.array([0.0, 0.0]) if pos is None else pos
        self.proj = geo.Projection()
        self.t = t


class SpeedCostParams(ParamBlock):
    DEFAULTS = dict(p_v=0.1, p_a=1.0)


class SpeedParams(ParamBlock):
    DEFAULTS = dict(horizon=250, step=0.5, ref_step=0.5, a_min=-2.5, a_max=2.5, j_min=-1.5, j_max=1.5, max_a_total=5.0,
                    max_lat_acc=1.5, max_traffic_light_stop_acc=2.0, d_lat_leader_safe=1.0, dt_safe=1.5, min_d_safe=1.0,
                    min_v_profile=1.0, time_constr_alpha=10.0, time_constr_beta=0.005, cost_func=SpeedCostParams)


class VelocityOptim:

    def __init__(self, solver=None):
        opt = self.opt = make_solver("velocity_profile_space", solver)
        opt.max_iterations = 20
        opt.lg_mult_limit = 0.1
        opt.barrier_weight[:] = 1000.0
        opt.lagrange_multiplier[:] = 0.0
        self.path_prev = None
        self.ss = self.shifts = np.zeros((1,))
        self.v_lim = self.v_opt = self.stop_mask = np.zeros((1,))
        self.v_ref = np.zeros((1, 2))                       # reference speed and its derivative per station
        self.s_leader, self.v_leader = 10**6, 0.0
        self.reset_counter = 0
        self.man_max_time_cons, self.man_min_time_cons = [], []

    # ---- stations of this cycle relative to the previous one ------------------------------------------------------
    def update_shifts(self, path, params):
        self.ss = np.arange(0.0, params.horizon * params.step, params.step)
        moved = 0.0 if self.path_prev is None else geo.project(self.path_prev[:, :2], path[0, :2]).arc_len
        self.shifts = self.ss.copy() if self.path_prev is None else self.ss + moved
        self.path_prev = path

    def shift_interp(self, arr, axis=0, interp_kind="linear"):
        """Values given on the previous cycle's stations, read at this cycle's."""
        return interp1d(self.ss, arr, kind=interp_kind, axis=axis, fill_value="extrapolate")(self.shifts)

    # ---- the vehicle to follow --------------------------------------------------------------------------------------
    def update_leader(self, path, env, params):
        """The tracked object closest ahead whose hull lies within the vehicle's swept width of the path."""
        self.s_leader, self.v_leader = NO_LEADER, 0.0
        veh = env.vehicle_state
        reach = veh.width / 2.0 + params.d_lat_leader_safe
        centre = path[:, :2]
        own_station = geo.project(centre, (veh.x, veh.y)).arc_len
        for obj in env.get_all_tracks():
            at_centre = geo.project(centre, obj.pos)
            if abs(at_centre.distance) - obj.hull_radius >= reach:          # cheap rejection by the bounding circle
                continue
            corners = geo.project(centre, np.asarray(obj.hull, dtype=np.float64).reshape(-1, 2))
            if not all(c.in_bounds for c in corners):
                continue
            offsets = np.array([c.distance for c in corners])
            one_sided = np.all(offsets >= 0.0) or np.all(offsets < 0.0)
            if one_sided and np.min(np.abs(offsets)) > reach:
                continue
            rear = np.min([c.arc_len for c in corners])
            if rear >= self.s_leader:
                continue
            self.s_leader = rear
            self.v_leader = max(0.0, obj.v * np.cos(at_centre.angle - obj.yaw))     # its speed along the path
            if self.v_leader > 0.5:
                self.s_leader -= own_station

    # ---- pieces of one cycle ------------------------------------------------------------------------------------------
    def _speed_limit(self, path, env, params, veh):
        limit = np.array(path[:, 5], dtype=np.float64)
        # behind the leader: its speed, scaled down the closer it is, from a time-gap distance on
        standstill_gap = veh.rear_axis_to_front + params.min_d_safe
        keep = self.v_leader * params.dt_safe + standstill_gap
        speed_ratio = min(4.0, self.v_leader / max(0.01, veh.v))
        closeness = self.s_leader / keep * speed_ratio
        geo.add_vel_constraint(limit, int((self.s_leader - keep) / self.opt.step), self.v_leader * closeness, length=20)
        for start, end, speed in getattr(env, "man_vel_cons", []):          # speed limits between two positions
            a, b = geo.project(path[:, :2], start), geo.project(path[:, :2], end)
            geo.add_vel_constraint(limit, a.index, speed, b.index - a.index, 0)
        return limit

    def _followable_reference(self, limit, fresh, params):
        n = self.opt.horizon
        if self.v_ref.shape[0] != n:                                          # horizon changed: keep the start values only
            resized = np.zeros((n, 2))
            resized[0] = self.v_ref[0]
            self.v_ref = resized
        if fresh:
            self.v_ref[0] = (limit[0], 0.0)
        else:
            self.v_ref = self.shift_interp(self.v_ref)
        v0, dv0 = self.v_ref[0]
        self.v_ref = rampify_profile(v0, dv0, limit, params.a_min, params.a_max, params.j_min, params.j_max,
                                     params.min_v_profile, self.opt.step)

    def _time_windows(self, path, env, params, veh):
        """Crossing-time windows of the manoeuvre module as per-station bounds on the solution's time state, plus a
        weight that lets the speed reference go near an earliest-time constraint."""
        opt, n = self.opt, self.opt.horizon
        latest, earliest = np.ones((n,)) * UNBOUNDED_TIME, np.zeros((n,))
        offset, weight = np.ones((n,)), np.ones((n,))
        here = geo.project(path[:, :2], np.array([veh.x, veh.y]))
        t_here = (1.0 - here.alpha) * opt.x[here.start, 1] + here.alpha * opt.x[here.end, 1]
        t_origin = env.t - t_here                                            # wall-clock time of station 0
        self.man_min_time_cons = [TimeConstr(pos=pos, t=lo) for pos, lo, hi in env.man_time_cons]
        self.man_max_time_cons = [TimeConstr(pos=pos, t=hi) for pos, lo, hi in env.man_time_cons]
        stations = np.arange(0, n) * opt.step
        for cons, is_earliest in [(c, True) for c in self.man_min_time_cons] + [(c, False) for c in self.man_max_time_cons]:
            cons.proj = geo.project(path[:, :2], cons.pos)
            i = cons.proj.index
            if i >= n - 1 or env.t > cons.t:                                   # beyond the horizon, or already over
                continue
            relative = cons.t - t_origin
            if is_earliest:
                earliest[i] = max(0.0, relative)
                offset[i] = relative - opt.x[i, 1]
                weight = np.minimum(weight, ((stations - (cons.proj.arc_len - params.time_constr_alpha))
                                             * params.time_constr_beta) ** 2)
            else:
                latest[i] = max(0.0, relative)
        q = opt.params
        q.ref_t_max, q.ref_t_min, q.ref_t_offset, q.ref_v_weight = latest, earliest, offset, weight
        return earliest

    # ---- one cycle ------------------------------------------------------------------------------------------------
    @runtime
    def update(self, path, env, params):
        veh = env.vehicle_state
        fresh = self.reset_counter != env.reset_counter
        self.reset_counter = env.reset_counter
        params.horizon = min(len(path), params.horizon)
        opt = self.opt
        opt.integrator_type = opt.EULER
        opt.horizon, opt.step = params.horizon, params.step
        opt.params.ref_step, opt.params.max_a_total = params.ref_step, params.max_a_total
        opt.u_max[:], opt.u_min[:] = params.a_max, params.a_min
        geo.merge(opt.params, params.cost_func)

        self.update_shifts(path, params)
        self.update_leader(path, env, params)
        # warm start on the new stations; the time state restarts at zero (the time windows are relative to it)
        opt.x[:-1] = self.shift_interp(np.array(opt.x[:-1, :]))
        opt.x[:, 1] -= opt.x[0, 1]
        opt.u = self.shift_interp(np.array(opt.u), interp_kind="zero")
        opt.lagrange_multiplier = self.shift_interp(np.array(opt.lagrange_multiplier))

        limit = self._speed_limit(path, env, params, veh)
        self._followable_reference(limit, fresh, params)
        if fresh:
            opt.x[0, 0], opt.x[0, 1] = veh.v, veh.a
        opt.params.ref_v, opt.params.ref_k = self.v_ref[:, 0], path[:, 4]
        earliest = self._time_windows(path, env, params, veh)
        opt.update()

        solution = np.asarray(opt.x)[:-1]
        moving = (earliest - solution[:, 1] <= 0.0) | (solution[:, 0] > params.min_v_profile * 1.1)
        self.stop_mask = geo.zero_after_first_zero(((limit >= params.min_v_profile) * moving).astype(float))
        self.v_lim = limit
        self.v_opt = solution[:, 0].copy() * self.stop_mask
