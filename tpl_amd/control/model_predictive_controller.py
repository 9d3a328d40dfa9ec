"""``ModelPredictiveController`` (library/tpl/control/model_predictive_controller.py): the trajectory-tracking MPC of
tplsim's default scenario -- HP-2's caller.  Per tick: the planner's trajectory resampled into the solver's reference
arrays, bounds and cost weights from the shared parameters, dead-time compensation by rolling the applied controls
through the model, projection on the reference, ``Optim.update()``, controls from stage 1, the predicted trajectory
for display.  ``solver`` defaults to the HIP solver of tpl_amd.optim (batch of one)."""
import copy

import numpy as np

from ..planning.base import Trajectory
from ..planning.geometry import project, resample_path
from ..planning.path_vel_decomp._common import make_solver, runtime
from ..store import merge


class BaseController:
    """control/base_controller.py"""

    def __init__(self, shared, lock_shared):
        self.shared = shared
        self.lock_shared = lock_shared

    def reinit_state(self):
        pass

    def reinit_params(self):
        pass

    def update(self, con_input):
        return (0.0, 0.0), Trajectory()


class IdleCompensationParams:
    """:12-21"""

    def __init__(self):
        self.active = False
        self.min_acc = -2.0
        self.jerk = -1.0
        self.veh_thresh = 0.5
        self.traj_thresh = 0.1
        self.traj_look_ahead_steps = 5


class CostFunctionParams:
    """:24-38"""

    def __init__(self):
        self.pd = 10.0
        self.pv = 5.0
        self.pdelta = 0.0
        self.min_pdelta_dot = 0.1
        self.pdelta_dot = 0.0
        self.min_p_phi_dot = 0.0
        self.p_phi_dot = 0.0
        self.p_phi = 0.01
        self.p_phi_ref_dot_diff = 0.01
        self.pa = 2.0
        self.pj = 0.5


class Params:
    """:41-59"""

    def __init__(self):
        self.horizon = 60
        self.step = 0.05
        self.max_iterations = 20
        self.acc_min = -3.0
        self.acc_max = 3.0
        self.jerk_min = -3.0
        self.jerk_max = 1.5
        self.steer_rate_min = -1.0
        self.steer_rate_max = 1.0
        self.ref_step = 0.5
        self.cost_function = CostFunctionParams()
        self.idle_comp = IdleCompensationParams()

    def deepcopy(self):
        return copy.deepcopy(self)


class ModelPredictiveController(BaseController):

    def __init__(self, shared, lock_shared, solver=None):
        self.shared = shared
        self.lock_shared = lock_shared
        self.opt = make_solver("trajectory_tracking_mpc", solver)                       # :67-74
        self.opt.integrator_type = self.opt.HEUN
        self.opt.lg_mult_limit = 0.0
        self.opt.barrier_weight[:] = 10000.0
        self.opt.lagrange_multiplier[:] = 0.0
        self.opt.params.v_ch = 32.0
        self.opt.params.max_delta = 0.7
        self.lat_dist_to_traj = 0.0                                                     # :77-88
        self.jerk = 0.0
        self.dead_time_trajectory = np.zeros((0, 6))
        self.controls = (0.0, 0.0)
        self.con_traj = Trajectory()
        self.last_update_time = 0.0
        self.ctrl_vars_history = []
        self.idle_comp_acc = 0.0
        self.idle_comp_steer = 0.0
        self.cycle_time_estimate = 0.01
        with self.lock_shared():                                                        # :91-92
            self.shared.params = Params()

    @runtime
    def update(self, con_input):
        t = con_input.t
        veh = con_input.vehicle
        traj = con_input.trajectory
        opt = self.opt
        delta_time = t - self.last_update_time                                          # :104-108
        if delta_time < 0.0 or delta_time > 1.0:
            self.ctrl_vars_history = []
        else:
            self.cycle_time_estimate = self.cycle_time_estimate * 0.95 + delta_time * 0.05
        with self.lock_shared():
            params = self.shared.params.deepcopy()
        if len(traj.time) < 2:
            return self.controls, self.con_traj
        traj_arr = np.vstack([traj.x, traj.y, traj.orientation, traj.s, traj.curvature, traj.velocity]).T   # :117-128
        traj = resample_path(traj_arr, params.ref_step, 100, start_index=0, zero_vel_at_end=True)
        if traj is None:
            return self.controls, self.con_traj
        opt.u_min[:, 0] = params.jerk_min                                               # :134-139
        opt.u_max[:, 0] = params.jerk_max
        opt.u_min[:, 1] = params.steer_rate_min
        opt.u_max[:, 1] = params.steer_rate_max
        opt.params.min_acc = params.acc_min
        opt.params.max_acc = params.acc_max
        merge(opt.params, params.cost_function)                                         # :142
        opt.horizon = params.horizon                                                    # :145-155
        opt.step = params.step
        opt.max_iterations = params.max_iterations
        opt.params.l = veh.wheel_base
        opt.params.ref_x = traj[:, 0]
        opt.params.ref_y = traj[:, 1]
        opt.params.ref_phi = traj[:, 2]
        opt.params.ref_k = traj[:, 4]
        opt.params.ref_v = traj[:, 5]
        opt.params.ref_step = params.ref_step
        opt.params.a_offset = 9.81 * np.sin(veh.pitch)
        self.lat_dist_to_traj = project(np.vstack((traj[:, 0], traj[:, 1])).T, (veh.x, veh.y)).distance   # :158-159
        x0 = np.array([veh.x, veh.y, veh.phi, veh.delta, veh.v, 0.0, veh.a])             # :163-171
        if veh.dead_time_steer > 0.0:                                                   # :173-184
            x0s = []
            rt = t
            dead_time_index = int(veh.dead_time_steer / self.cycle_time_estimate + 1e-5)
            for acc, delta in self.ctrl_vars_history[-dead_time_index:]:
                x0s.append(np.array([rt, *x0]))
                u = np.array([0.0, 0.0])
                x0[3] = delta
                x0[6] = acc
                x0 = np.array(opt.dynamics(x0, u, 0, self.cycle_time_estimate))
                rt += self.cycle_time_estimate
            x0s.append(np.array([rt, *x0]))
            self.dead_time_trajectory = np.array(x0s)
        proj = project(traj[:, :2], x0[:2])                                             # :187-188
        x0[5] = proj.arc_len
        opt.x[0] = x0                                                                   # :191-192
        opt.update()
        j = 1.0                                                                         # :195-199
        steering_angle = (1.0 - j) * opt.x[0][3] + j * opt.x[1][3]
        steering_angle = min(veh.delta_max, max(-veh.delta_max, steering_angle))
        acc = (1.0 - j) * opt.x[0][6] + j * opt.x[1][6]
        acc = min(params.acc_max, max(params.acc_min, acc))
        if delta_time > 0.0:                                                            # :202-205
            self.ctrl_vars_history.append((acc, steering_angle))
        if len(self.ctrl_vars_history) > 100:
            self.ctrl_vars_history.pop(0)
        self.last_update_time = t
        vel_idx = 5                                                                     # :211-220: idle compensation
        if params.idle_comp.active and (veh.v < params.idle_comp.veh_thresh
                                        and traj[params.idle_comp.traj_look_ahead_steps, vel_idx] < params.idle_comp.traj_thresh):
            self.idle_comp_acc += params.idle_comp.jerk * delta_time
            steering_angle = self.idle_comp_steer
        else:
            self.idle_comp_steer = steering_angle
            self.idle_comp_acc = 0.0
        self.idle_comp_acc = min(0.0, max(params.idle_comp.min_acc, self.idle_comp_acc))
        acc += self.idle_comp_acc
        self.jerk = opt.u[0][0]
        self.controls = (acc, steering_angle)
        self.con_traj = Trajectory()                                                    # :226-235
        self.con_traj.time = t + np.arange(0, opt.T * opt.dt, opt.dt)
        self.con_traj.x = opt.x[:-1, 0]
        self.con_traj.y = opt.x[:-1, 1]
        self.con_traj.orientation = opt.x[:-1, 2]
        self.con_traj.velocity = opt.x[:-1, 4]
        self.con_traj.curvature = np.tan(opt.x[:-1, 3]) / veh.wheel_base
        self.con_traj.s = opt.x[:-1, 5] - opt.x[0, 5]
        self.con_traj.acceleration = opt.u[:, 0]
        return self.controls, self.con_traj
