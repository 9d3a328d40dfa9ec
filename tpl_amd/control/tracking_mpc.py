"""The trajectory-tracking MPC plugins around the batched iLQR solver -- HP-2's callers in tplsim.

Behaviour of ``tpl.control.ModelPredictiveController`` (reference indexed by arc length, problem
``trajectory_tracking_mpc``, model_predictive_controller.py:62-237) and ``ModelPredictiveControllerTime`` (reference
indexed by time, ``trajectory_tracking_mpc_time``, model_predictive_controller_time.py:65-218), built here as ONE tick
pipeline with two small specialisations:

    shared parameters -> reference arrays -> solver bounds / weights / horizon -> initial state
    -> roll the controls applied during the actuator dead time through the model -> solve
    -> controls from the first predicted stage -> idle compensation -> predicted trajectory for display

``solver`` defaults to the HIP solver of tpl_amd.optim (a batch of one); tests hand in the CPU restatement."""
import numpy as np

from ..planning._kit import ParamBlock, fill_trajectory
from ..planning.base import Trajectory
from ..planning.geometry import project, resample_path
from ..planning.path_vel_decomp._common import make_solver, runtime
from ..store import merge

GRAVITY = 9.81
HISTORY_LIMIT = 100                     # applied (acceleration, steering angle) pairs kept for the dead-time roll-out


class BaseController:
    """tpl.control.BaseController: what tplsim calls on a controller plugin."""

    def __init__(self, shared, lock_shared):
        self.shared, self.lock_shared = shared, lock_shared

    def reinit_state(self):
        pass

    def reinit_params(self):
        pass

    def update(self, con_input):
        return (0.0, 0.0), Trajectory()


class IdleCompensationParams(ParamBlock):
    DEFAULTS = dict(active=False, min_acc=-2.0, jerk=-1.0, veh_thresh=0.5, traj_thresh=0.1, traj_look_ahead_steps=5)


class CostFunctionParams(ParamBlock):
    DEFAULTS = dict(pd=10.0, pv=5.0, pdelta=0.0, min_pdelta_dot=0.1, pdelta_dot=0.0, min_p_phi_dot=0.0, p_phi_dot=0.0,
                    p_phi=0.01, p_phi_ref_dot_diff=0.01, pa=2.0, pj=0.5)


class Params(ParamBlock):
    """Shared parameters of the arc-length controller."""
    DEFAULTS = dict(horizon=60, step=0.05, max_iterations=20, acc_min=-3.0, acc_max=3.0, jerk_min=-3.0, jerk_max=1.5,
                    steer_rate_min=-1.0, steer_rate_max=1.0, ref_step=0.5, cost_function=CostFunctionParams,
                    idle_comp=IdleCompensationParams)


class TimeCostFunctionParams(ParamBlock):
    DEFAULTS = dict(CostFunctionParams.DEFAULTS, pdelta_dot=0.1, p_phi=0.0, p_phi_ref_dot_diff=0.0)


class TimeParams(ParamBlock):
    """Shared parameters of the time-indexed controller."""
    DEFAULTS = dict(horizon=40, step=0.05, max_iterations=20, cycle_time=0.01, acc_min=-3.0, acc_max=3.0, jerk_min=-3.0,
                    jerk_max=1.5, steer_rate_min=-1.0, steer_rate_max=1.0, cog_pos=0.5, ref_dt=0.1,
                    cost_function=TimeCostFunctionParams, idle_comp=IdleCompensationParams)


def _clamp(value, lo, hi):
    return min(hi, max(lo, value))


class _TrackingMpc(BaseController):
    """One tick of either controller; the subclasses say which problem, which state layout and how the reference and
    the initial state are built."""
    PROBLEM = None
    PARAM_CLASS = None
    STEER_STATE, SPEED_STATE, ACC_STATE = 3, 4, None        # positions of delta, v, a in the model's state vector

    def __init__(self, shared, lock_shared, solver=None):
        super().__init__(shared, lock_shared)
        opt = self.opt = make_solver(self.PROBLEM, solver)
        opt.integrator_type = opt.HEUN
        opt.lg_mult_limit = 0.0
        opt.barrier_weight[:] = 10000.0
        opt.lagrange_multiplier[:] = 0.0
        opt.params.v_ch, opt.params.max_delta = 32.0, 0.7
        self.lat_dist_to_traj = 0.0
        self.jerk = 0.0
        self.dead_time_trajectory = np.zeros((0, self.DEAD_TIME_COLUMNS))
        self.controls = (0.0, 0.0)
        self.con_traj = Trajectory()
        self.last_update_time = 0.0
        self.ctrl_vars_history = []
        self.idle_comp_acc = self.idle_comp_steer = 0.0
        with self.lock_shared():
            self.shared.params = self.PARAM_CLASS()

    # -- specialisation points ---------------------------------------------------------------------------------
    def _note_elapsed(self, elapsed):
        raise NotImplementedError

    def _reference(self, traj, p):
        raise NotImplementedError

    def _load_reference(self, ref, veh, p):
        raise NotImplementedError

    def _state_of(self, veh, p):
        raise NotImplementedError

    def _roll_step(self, p):
        raise NotImplementedError

    def _anchor(self, x0, ref):
        return x0

    def _prediction_rows(self, veh):
        raise NotImplementedError

    # -- the tick ------------------------------------------------------------------------------------------------
    @runtime
    def update(self, con_input):
        now, veh, plan = con_input.t, con_input.vehicle, con_input.trajectory
        elapsed = now - self.last_update_time
        self._note_elapsed(elapsed)
        with self.lock_shared():
            p = self.shared.params.deepcopy()
        ref = self._reference(plan, p) if len(plan.time) >= 2 else None
        if ref is None:
            return self.controls, self.con_traj                 # nothing to track: keep the last answer

        self._load_bounds_and_weights(p, veh)
        self._load_reference(ref, veh, p)
        x0 = self._through_dead_time(self._state_of(veh, p), now, veh, p)
        self.opt.x[0] = self._anchor(x0, ref)
        self.opt.update()

        acc, steer = self._first_stage_controls(veh, p)
        if elapsed > 0.0:
            self.ctrl_vars_history.append((acc, steer))
        del self.ctrl_vars_history[:max(0, len(self.ctrl_vars_history) - HISTORY_LIMIT)]
        self.last_update_time = now
        acc, steer = self._idle_compensation(acc, steer, elapsed, veh, ref, p.idle_comp)

        self.jerk = self.opt.u[0][0]
        self.controls = (acc, steer)
        self.con_traj = self._prediction(now, veh)
        return self.controls, self.con_traj

    def _load_bounds_and_weights(self, p, veh):
        opt = self.opt
        for col, (lo, hi) in enumerate(((p.jerk_min, p.jerk_max), (p.steer_rate_min, p.steer_rate_max))):
            opt.u_min[:, col] = lo
            opt.u_max[:, col] = hi
        opt.params.min_acc, opt.params.max_acc = p.acc_min, p.acc_max
        merge(opt.params, p.cost_function)                      # every weight the problem knows
        opt.horizon, opt.step, opt.max_iterations = p.horizon, p.step, p.max_iterations
        opt.params.l = veh.wheel_base
        opt.params.a_offset = GRAVITY * np.sin(veh.pitch)       # slope compensation

    def _through_dead_time(self, x0, now, veh, p):
        """The commands of the last ``dead_time_steer`` seconds have not reached the actuators yet: apply them to the
        model, one cycle each, and plan from where that leaves the vehicle.  Keeps the roll-out for display."""
        if veh.dead_time_steer <= 0.0:
            return x0
        step = self._roll_step(p)
        pending = self.ctrl_vars_history[-int(veh.dead_time_steer / step + 1e-5):]
        rows, no_input, clock = [], np.zeros(2), now
        for acc, steer in pending:
            rows.append(np.concatenate(([clock], x0)))
            x0[self.STEER_STATE], x0[self.ACC_STATE] = steer, acc
            x0 = np.array(self.opt.dynamics(x0, no_input, 0, step))
            clock += step
        rows.append(np.concatenate(([clock], x0)))
        self.dead_time_trajectory = np.array(rows)
        return x0

    def _first_stage_controls(self, veh, p):
        nxt = self.opt.x[1]                                     # the state one solver step ahead carries the commands
        return (_clamp(nxt[self.ACC_STATE], p.acc_min, p.acc_max),
                _clamp(nxt[self.STEER_STATE], -veh.delta_max, veh.delta_max))

    def _idle_compensation(self, acc, steer, elapsed, veh, ref, ic):
        """Standing still on a plan that wants to stand still: ramp in extra braking and freeze the steering."""
        holding = ic.active and veh.v < ic.veh_thresh and ref[ic.traj_look_ahead_steps, 5] < ic.traj_thresh
        if holding:
            self.idle_comp_acc += ic.jerk * elapsed
            steer = self.idle_comp_steer
        else:
            self.idle_comp_steer, self.idle_comp_acc = steer, 0.0
        self.idle_comp_acc = _clamp(self.idle_comp_acc, ic.min_acc, 0.0)
        return acc + self.idle_comp_acc, steer

    def _prediction(self, now, veh):
        opt = self.opt
        out = Trajectory()
        out.time = now + np.arange(0, opt.T * opt.dt, opt.dt)
        out.x, out.y, out.orientation = opt.x[:-1, 0], opt.x[:-1, 1], opt.x[:-1, 2]
        out.velocity = opt.x[:-1, self.SPEED_STATE]
        out.curvature = np.tan(opt.x[:-1, self.STEER_STATE]) / veh.wheel_base
        self._prediction_extras(out)
        return out


class ModelPredictiveController(_TrackingMpc):
    """Arc-length reference: state (x, y, phi, delta, v, s_ref, a); the plan is resampled every ``ref_step`` metres and
    the state's s_ref is the vehicle's projection on it."""
    PROBLEM, PARAM_CLASS = "trajectory_tracking_mpc", Params
    ACC_STATE, ARC_STATE, DEAD_TIME_COLUMNS = 6, 5, 6
    REFERENCE_POINTS = 100

    def __init__(self, shared, lock_shared, solver=None):
        super().__init__(shared, lock_shared, solver)
        self.cycle_time_estimate = 0.01                         # low-passed time between ticks

    def _note_elapsed(self, elapsed):
        if 0.0 <= elapsed <= 1.0:
            self.cycle_time_estimate = 0.95 * self.cycle_time_estimate + 0.05 * elapsed
        else:
            self.ctrl_vars_history = []                         # time jumped: the history says nothing any more

    def _reference(self, plan, p):
        columns = np.vstack([plan.x, plan.y, plan.orientation, plan.s, plan.curvature, plan.velocity]).T
        return resample_path(columns, p.ref_step, self.REFERENCE_POINTS, start_index=0, zero_vel_at_end=True)

    def _load_reference(self, ref, veh, p):
        q = self.opt.params
        q.ref_x, q.ref_y, q.ref_phi, q.ref_k, q.ref_v = ref[:, 0], ref[:, 1], ref[:, 2], ref[:, 4], ref[:, 5]
        q.ref_step = p.ref_step
        self.lat_dist_to_traj = project(ref[:, :2], (veh.x, veh.y)).distance

    def _state_of(self, veh, p):
        return np.array([veh.x, veh.y, veh.phi, veh.delta, veh.v, 0.0, veh.a])

    def _roll_step(self, p):
        return self.cycle_time_estimate

    def _anchor(self, x0, ref):
        x0[self.ARC_STATE] = project(ref[:, :2], x0[:2]).arc_len
        return x0

    def _prediction_extras(self, out):
        arc = self.opt.x[:, self.ARC_STATE]
        out.s = arc[:-1] - arc[0]
        out.acceleration = self.opt.u[:, 0]


class ModelPredictiveControllerTime(_TrackingMpc):
    """Time-indexed reference: state (x, y, phi, delta, v, a) of the point ``cog_pos`` wheel bases ahead of the rear
    axle; the plan's samples (``ref_dt`` apart) are the reference as they are."""
    PROBLEM, PARAM_CLASS = "trajectory_tracking_mpc_time", TimeParams
    ACC_STATE, DEAD_TIME_COLUMNS = 5, 5

    def _note_elapsed(self, elapsed):
        if elapsed < 0.0:
            self.ctrl_vars_history = []

    def _reference(self, plan, p):
        return plan

    def _load_reference(self, plan, veh, p):
        q = self.opt.params
        q.ref_x, q.ref_y, q.ref_phi, q.ref_v = plan.x, plan.y, plan.orientation, plan.velocity
        q.ref_dt, q.ref_t_offset, q.cog_pos = p.ref_dt, veh.dead_time_steer, p.cog_pos
        self.lat_dist_to_traj = project(np.vstack((plan.x, plan.y)).T, (veh.x, veh.y)).distance

    def _state_of(self, veh, p):
        ahead = p.cog_pos * veh.wheel_base
        return np.array([veh.x + np.cos(veh.phi) * ahead, veh.y + np.sin(veh.phi) * ahead, veh.phi, veh.delta, veh.v, veh.a])

    def _roll_step(self, p):
        return p.cycle_time

    def _prediction_extras(self, out):
        out.acceleration = self.opt.x[:, self.ACC_STATE]
