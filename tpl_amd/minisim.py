"""Mini-tplsim: the closed loop of BASELINE configs[0] ("tplsim --scenario default, OCP solver, N=20 horizon,
single vehicle") reduced to what exercises the hot path -- the simulator's ego model and the model-predictive
controller's per-tick use of the genopt solver.  No planner, map or IPC: the plan is a straight lane at constant
speed.  Pure host arithmetic around ``Optim.update()``; works with any object that has the reference's ``Optim``
surface (the HIP solver of tpl_amd.optim, or -- in tests and bench.py's CPU leg -- the CPU oracle).

Restated from the reference (no code shared):
  * ``EgoSim.step``     SimCore.update_ego, library/tpl/simulation/core.py:91-135: actuator dead time through
                        (time, value) buffers, kinematic single-track model with characteristic velocity,
                        explicit Euler with the simulator's integration step.
  * ``MpcShim.update``  ModelPredictiveController.update, library/tpl/control/model_predictive_controller.py:96-237:
                        bounds / parameters / reference arrays written into the solver's views, dead-time
                        compensation by rolling the last applied controls through ``opt.dynamics``, projection on the
                        plan, ``opt.update()``, controls read from stage 1 of the solution.
"""
import math
import time

import numpy as np


class EgoSim:
    """Ego vehicle of the simulator (core.py:91-135) with the default scenario's vehicle constants."""

    def __init__(self, x=0.0, y=0.4, yaw=0.05, v=6.0, wheel_base=3.165, v_ch=32.0, acc_dead_time=0.1,
                 steer_dead_time=0.1, max_v=40.0, min_v=0.0, max_steer_angle=0.7):
        self.x, self.y, self.yaw, self.v = x, y, yaw, v
        self.a = 0.0
        self.steer_angle = 0.0
        self.control_acc = 0.0
        self.control_steer = 0.0
        self.wheel_base, self.v_ch = wheel_base, v_ch
        self.acc_dead_time, self.steer_dead_time = acc_dead_time, steer_dead_time
        self.max_v, self.min_v, self.max_steer_angle = max_v, min_v, max_steer_angle
        self._acc_buf = []
        self._steer_buf = []

    def step(self, t, dt):
        if dt > 0.0:
            self._acc_buf.append((t, self.control_acc))
            self._steer_buf.append((t, self.control_steer))
            while len(self._acc_buf) > self.acc_dead_time // dt + 1:
                self._acc_buf.pop(0)
            while len(self._steer_buf) > self.steer_dead_time // dt + 1:
                self._steer_buf.pop(0)
        if self.acc_dead_time == 0.0 and self._acc_buf:
            self.a = self._acc_buf[-1][1]
        else:
            for at, ac in self._acc_buf:
                if t - at <= self.acc_dead_time:
                    self.a = ac
                    break
        if self.steer_dead_time == 0.0 and self._steer_buf:
            self.steer_angle = self._steer_buf[-1][1]
        else:
            for st, sc in self._steer_buf:
                if t - st <= self.steer_dead_time:
                    self.steer_angle = sc
                    break
        self.x += dt * self.v * math.cos(self.yaw)
        self.y += dt * self.v * math.sin(self.yaw)
        self.yaw += dt * self.v / (self.wheel_base * (1.0 + (self.v / self.v_ch) ** 2)) * math.tan(self.steer_angle)
        self.yaw = (self.yaw + math.pi) % (2.0 * math.pi) - math.pi
        self.v += dt * self.a
        self.v = min(self.max_v, max(self.min_v, self.v))
        self.steer_angle = min(self.max_steer_angle, max(-self.max_steer_angle, self.steer_angle))


class MpcShim:
    """The controller's per-tick solver use (model_predictive_controller.py:62-237) on a straight plan along the
    x axis with speed ``ref_v``; cost weights of data/params/control/default/state.json."""

    def __init__(self, opt, horizon=20, step=0.05, max_iterations=20, ref_v=8.0, ref_step=0.5, n_ref=100,
                 acc_min=-3.0, acc_max=3.0, jerk_min=-3.0, jerk_max=1.5, steer_rate_min=-1.0, steer_rate_max=1.0):
        self.opt = opt
        self.horizon, self.step, self.max_iterations = horizon, step, max_iterations
        self.acc_min, self.acc_max = acc_min, acc_max
        self.jerk = (jerk_min, jerk_max)
        self.steer_rate = (steer_rate_min, steer_rate_max)
        self.ref_step, self.n_ref, self.ref_v = ref_step, n_ref, ref_v
        opt.integrator_type = opt.HEUN                      # :66-69
        opt.lg_mult_limit = 0.0
        opt.barrier_weight[:] = 10000.0
        opt.lagrange_multiplier[:] = 0.0
        opt.params.v_ch = 32.0                              # :71-72
        opt.params.max_delta = 0.7
        for k, v in dict(pd=10.0, pv=5.0, pdelta=0.0, min_pdelta_dot=0.1, pdelta_dot=0.0, min_p_phi_dot=0.0,
                         p_phi_dot=0.05, p_phi=1000.0, p_phi_ref_dot_diff=0.01, pa=2.0, pj=0.5).items():
            setattr(opt.params, k, v)
        self.ctrl_vars_history = []
        self.cycle_time_estimate = 0.01
        self.last_update_time = 0.0
        self.controls = (0.0, 0.0)
        self.solve_ms = []

    def update(self, t, veh, dead_time_steer):
        opt = self.opt
        delta_time = t - self.last_update_time                                        # :104-108
        if delta_time < 0.0 or delta_time > 1.0:
            self.ctrl_vars_history = []
        else:
            self.cycle_time_estimate = self.cycle_time_estimate * 0.95 + delta_time * 0.05
        # the plan, resampled from the vehicle's projection onwards (:117-131): straight lane, constant speed,
        # zero velocity at the end like resample_path(zero_vel_at_end=True)
        s0 = max(0.0, veh.x)
        s = s0 + np.arange(self.n_ref) * self.ref_step
        ref_v = np.full(self.n_ref, self.ref_v)
        ref_v[-1] = 0.0
        opt.u_min[:, 0], opt.u_max[:, 0] = self.jerk                                    # :134-139
        opt.u_min[:, 1], opt.u_max[:, 1] = self.steer_rate
        opt.params.min_acc, opt.params.max_acc = self.acc_min, self.acc_max
        opt.horizon = self.horizon                                                      # :145-155
        opt.step = self.step
        opt.max_iterations = self.max_iterations
        opt.params.l = veh.wheel_base
        opt.params.ref_x = s
        opt.params.ref_y = np.zeros(self.n_ref)
        opt.params.ref_phi = np.zeros(self.n_ref)
        opt.params.ref_k = np.zeros(self.n_ref)
        opt.params.ref_v = ref_v
        opt.params.ref_step = self.ref_step
        opt.params.a_offset = 0.0
        x0 = np.array([veh.x, veh.y, veh.yaw, veh.steer_angle, veh.v, 0.0, veh.a])      # :163-171
        if dead_time_steer > 0.0:                                                       # :173-184
            n = int(dead_time_steer / self.cycle_time_estimate + 1e-5)
            for acc, delta in (self.ctrl_vars_history[-n:] if n > 0 else []):
                x0[3] = delta
                x0[6] = acc
                x0 = np.array(opt.dynamics(x0, np.zeros(2), 0, self.cycle_time_estimate))
        x0[5] = min(max(x0[0] - s0, 0.0), (self.n_ref - 1) * self.ref_step)             # :187-188 (projection on a straight line)
        opt.x[0] = x0
        t0 = time.perf_counter()
        opt.update()                                                                    # :192
        self.solve_ms.append((time.perf_counter() - t0) * 1e3)
        steering_angle = min(veh.max_steer_angle, max(-veh.max_steer_angle, float(opt.x[1][3])))   # :195-199 (j = 1)
        acc = min(self.acc_max, max(self.acc_min, float(opt.x[1][6])))
        if delta_time > 0.0:                                                            # :202-205
            self.ctrl_vars_history.append((acc, steering_angle))
        if len(self.ctrl_vars_history) > 100:
            self.ctrl_vars_history.pop(0)
        self.last_update_time = t
        self.controls = (acc, steering_angle)
        return self.controls


def closed_loop(opt, ticks=1000, dt=0.01, int_step=0.005, **ego_kw):
    """``ticks`` controller cycles of ``dt`` (the default scenario's fixed_time_step 0.01 s, integration step
    0.005 s): returns the per-tick controls, the ego trace (x, y, yaw, v) and the controller."""
    ego = EgoSim(**ego_kw)
    con = MpcShim(opt)
    controls = np.zeros((ticks, 2))
    trace = np.zeros((ticks, 4))
    t = 0.0
    sub = max(1, int(round(dt / int_step)))
    for k in range(ticks):
        acc, steer = con.update(t, ego, ego.steer_dead_time)
        ego.control_acc, ego.control_steer = acc, steer
        for _ in range(sub):
            ego.step(t, dt / sub)
            t += dt / sub
        controls[k] = (acc, steer)
        trace[k] = (ego.x, ego.y, ego.yaw, ego.v)
    return controls, trace, con
