"""``PathVelDecompPlanner``: tplsim's default planner plugin on the MI355X solvers (behaviour of
planning/path_vel_decomp/path_vel_decomp_planner.py:34-161).  A cycle is: lateral stage (obstacle-avoiding offset
profile, or plain smoothing of the reference line) -> speed stage on the resulting path -> trajectory; three batched-iLQR
problems exist per planner, two are solved per cycle."""
import numpy as np

from ...store import Bundle
from .._kit import EnvSnapshot, ParamBlock
from ..base import BasePlanner, Trajectory
from ._common import runtime
from .lateral import OptimParams, PathOptim, PathSmoothing, SmoothingParams
from .speed import SpeedParams, VelocityOptim


class Params(ParamBlock):
    """``smooth_only`` switches obstacle avoidance off (the path is only smoothed); ``write_debug_data`` publishes the
    corridor and reference arrays every cycle."""
    DEFAULTS = dict(horizon=250, smooth_only=False, write_debug_data=False, path_smoothing=SmoothingParams,
                    path_optim=OptimParams, velocity_optim=SpeedParams)


def _offset_curve(line, offsets):
    """The reference line moved sideways by ``offsets`` (positive to the left)."""
    moved = line[:, :2].copy()
    moved[:, 0] -= np.sin(line[:, 2]) * offsets
    moved[:, 1] += np.cos(line[:, 2]) * offsets
    return moved


class PathVelDecompPlanner(BasePlanner):

    def __init__(self, shared, lock_shared, solvers=None):
        """``solvers``: optional {problem name: solver object} (tests); default: the HIP solvers."""
        pick = (solvers or {}).get
        self.shared, self.lock_shared = shared, lock_shared
        self.path_smoothing = PathSmoothing(pick("ref_line_smoother_k"))
        self.path_optim = PathOptim(pick("lateral_profile"))
        self.velocity_optim = VelocityOptim(pick("velocity_profile_space"))
        self.trajectory = Trajectory()
        self.env = EnvSnapshot()
        with self.lock_shared():
            self.shared.params = Params()
            self.shared.debug = Bundle()

    shift_path = staticmethod(_offset_curve)

    @runtime
    def update(self, sh_env):
        with self.lock_shared():
            params = self.shared.params.deepcopy()
        params.path_optim.horizon = params.velocity_optim.horizon = params.horizon
        env = self.env
        with sh_env.lock():
            if sh_env.local_map is None:
                return self.trajectory
            env.take(sh_env)
            self._ask_map_module(sh_env.local_map, env.vehicle_state, params.velocity_optim)

        active, idle, stage_params = ((self.path_smoothing, self.path_optim, params.path_smoothing) if params.smooth_only
                                      else (self.path_optim, self.path_smoothing, params.path_optim))
        idle.reset_required = True                               # the stage not in use restarts when it is switched on
        active.update(env, stage_params)
        path = active.opt_path
        speed = self.velocity_optim
        speed.update(path, env, params.velocity_optim)
        self.write_debug_data(env, params)

        out, sp = self.trajectory, params.velocity_optim
        out.time = env.t + speed.opt.x[:-1, 1].copy()            # the speed problem carries time as a state
        out.s = np.arange(0.0, sp.horizon * sp.step, sp.step)
        out.x, out.y, out.orientation, out.curvature = (path[:, c].copy() for c in (0, 1, 2, 4))
        out.velocity, out.acceleration = speed.v_opt, speed.opt.u.copy()
        return out

    @staticmethod
    def _ask_map_module(shared_map, veh, speed_params):
        """Requests the map module reads next cycle: keep the intersection paths current, move the window one step
        at a time, and start the speed-limit look-up a vehicle front + safety gap behind the window start."""
        shared_map.update_inters_paths = True
        shared_map.step_shift_idx = 1
        shared_map.shift_vel_lim = -int((veh.rear_axis_to_front + speed_params.min_d_safe) / shared_map.step_size_ref)

    def write_debug_data(self, env, params):
        speed, lateral = self.velocity_optim, self.path_optim
        with self.lock_shared():
            dbg = self.shared.debug = Bundle()
            dbg.s_leader, dbg.v_leader = speed.s_leader, speed.v_leader
            if not params.write_debug_data:
                return
            line = env.local_map.path[:min(params.horizon, env.local_map.steps_ref)].copy()
            q = lateral.opt.params
            dbg.opt_path = lateral.opt_path
            for name, offsets in (("lower_ref", q.d_lower_constr), ("upper_ref", q.d_upper_constr),
                                  ("upper_constr", lateral.d_upper_constr), ("lower_constr", lateral.d_lower_constr)):
                setattr(dbg, "d_" + name, offsets)
                setattr(dbg, "path_" + name, _offset_curve(line, offsets))
            dbg.d_offset, dbg.d_opt = q.d_offset, lateral.opt.x[:-1, 0]
            dbg.v_ref_weight = speed.opt.params.ref_v_weight
            dbg.man_min_time_cons, dbg.man_max_time_cons = speed.man_min_time_cons, speed.man_max_time_cons
            dbg.v_lim = np.maximum(0.0, speed.v_lim)
            dbg.v_ref = np.maximum(0.0, np.minimum(speed.v_lim, speed.v_ref[:, 0]))
