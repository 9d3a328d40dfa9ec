"""Mini-tplsim: the closed loop of BASELINE configs[0] ("tplsim --scenario default, OCP solver, N=20 horizon, single
vehicle") reduced to what exercises the hot path -- the simulator's ego vehicle under the trajectory-tracking MPC plugin
(tpl_amd.control.ModelPredictiveController), with a straight constant-speed plan instead of a planner.  Pure host
arithmetic around ``Optim.update()``; works with any solver object that has the reference's ``Optim`` surface (the HIP
solver, or -- in tests and bench.py's CPU leg -- the CPU restatement).

``EgoSim`` reproduces the behaviour of the simulator's ego update (library/tpl/simulation/core.py:91-135): actuator
commands take effect after a dead time, the vehicle is a kinematic single-track model whose yaw rate is attenuated by the
characteristic velocity, integrated with explicit Euler at the simulator's sub-step."""
import math
import time
from contextlib import contextmanager
from types import SimpleNamespace

import numpy as np

# the controller weights tpl ships for its default vehicle (data/params/control/default/state.json)
DEFAULT_VEHICLE_WEIGHTS = dict(pd=10.0, pv=5.0, pdelta=0.0, min_pdelta_dot=0.1, pdelta_dot=0.0, min_p_phi_dot=0.0,
                               p_phi_dot=0.05, p_phi=1000.0, p_phi_ref_dot_diff=0.01, pa=2.0, pj=0.5)


class _DelayLine:
    """A command that reaches its actuator ``dead_time`` seconds after it was issued."""

    def __init__(self, dead_time):
        self.dead_time = dead_time
        self.samples = []                                   # (time issued, value), oldest first

    def push(self, t, value, dt):
        self.samples.append((t, value))
        del self.samples[:max(0, len(self.samples) - int(self.dead_time // dt + 1))]

    def effective(self, t, current):
        if not self.samples:
            return current
        if self.dead_time == 0.0:
            return self.samples[-1][1]
        # the oldest command that is not older than the dead time; none -> the actuator keeps what it has
        return next((v for ts, v in self.samples if t - ts <= self.dead_time), current)


class EgoSim:
    """The simulator's ego vehicle with the default scenario's constants."""

    def __init__(self, x=0.0, y=0.4, yaw=0.05, v=6.0, wheel_base=3.165, v_ch=32.0, acc_dead_time=0.1,
                 steer_dead_time=0.1, max_v=40.0, min_v=0.0, max_steer_angle=0.7):
        self.x, self.y, self.yaw, self.v = x, y, yaw, v
        self.a = self.steer_angle = 0.0
        self.control_acc = self.control_steer = 0.0
        self.wheel_base, self.v_ch = wheel_base, v_ch
        self.acc_dead_time, self.steer_dead_time = acc_dead_time, steer_dead_time
        self.max_v, self.min_v, self.max_steer_angle = max_v, min_v, max_steer_angle
        self._acc_line, self._steer_line = _DelayLine(acc_dead_time), _DelayLine(steer_dead_time)

    def step(self, t, dt):
        if dt > 0.0:
            self._acc_line.push(t, self.control_acc, dt)
            self._steer_line.push(t, self.control_steer, dt)
        self.a = self._acc_line.effective(t, self.a)
        self.steer_angle = self._steer_line.effective(t, self.steer_angle)
        heading, speed = self.yaw, self.v
        self.x += dt * speed * math.cos(heading)
        self.y += dt * speed * math.sin(heading)
        yaw_rate = speed / (self.wheel_base * (1.0 + (speed / self.v_ch) ** 2)) * math.tan(self.steer_angle)
        self.yaw = (heading + dt * yaw_rate + math.pi) % (2.0 * math.pi) - math.pi
        self.v = min(self.max_v, max(self.min_v, speed + dt * self.a))
        self.steer_angle = min(self.max_steer_angle, max(-self.max_steer_angle, self.steer_angle))


def straight_plan(speed=8.0, length=400.0, step=0.5):
    """A planner trajectory along the x axis at constant speed."""
    s = np.arange(0.0, length, step)
    zeros = np.zeros_like(s)
    return SimpleNamespace(time=s / speed, s=s, x=s.copy(), y=zeros, orientation=zeros.copy(), curvature=zeros.copy(),
                           velocity=np.full_like(s, speed), acceleration=zeros.copy(), emergency=False)


@contextmanager
def _no_lock():
    yield


def closed_loop(opt, ticks=1000, dt=0.01, int_step=0.005, horizon=None, **ego_kw):
    """``ticks`` controller cycles of ``dt`` (the default scenario's fixed_time_step 0.01 s, integration sub-step
    0.005 s) of the MPC plugin on solver ``opt``: returns the per-tick controls, the ego trace (x, y, yaw, v) and the
    controller, whose ``solve_ms`` lists the wall time of every ``Optim.update()`` and ``tick_ms`` that of every whole
    ``ModelPredictiveController.update`` (reference look-up, dead-time prediction, bounds and weights, the solve, the
    read-out)."""
    from tpl_amd.control import ModelPredictiveController
    ego = EgoSim(**ego_kw)
    shared = SimpleNamespace()
    con = ModelPredictiveController(shared, _no_lock, solver=opt)
    shared.params.horizon = int(horizon if horizon is not None else opt.horizon)
    for name, weight in DEFAULT_VEHICLE_WEIGHTS.items():
        setattr(shared.params.cost_function, name, weight)
    con.solve_ms, con.tick_ms = [], []
    solve = opt.update

    def timed_update():
        t0 = time.perf_counter()
        solve()
        con.solve_ms.append((time.perf_counter() - t0) * 1e3)
    con.opt = _TimedSolver(opt, timed_update)
    plan = straight_plan()
    controls, trace = np.zeros((ticks, 2)), np.zeros((ticks, 4))
    sub = max(1, int(round(dt / int_step)))
    t = 0.0
    for k in range(ticks):
        veh = SimpleNamespace(x=ego.x, y=ego.y, phi=ego.yaw, delta=ego.steer_angle, v=ego.v, a=ego.a, pitch=0.0,
                              wheel_base=ego.wheel_base, delta_max=ego.max_steer_angle, dead_time_steer=ego.steer_dead_time)
        t_tick = time.perf_counter()
        (acc, steer), _ = con.update(SimpleNamespace(t=t, vehicle=veh, trajectory=plan))
        con.tick_ms.append((time.perf_counter() - t_tick) * 1e3)
        ego.control_acc, ego.control_steer = acc, steer
        for _ in range(sub):
            ego.step(t, dt / sub)
            t += dt / sub
        controls[k] = (acc, steer)
        trace[k] = (ego.x, ego.y, ego.yaw, ego.v)
    return controls, trace, con


class _TimedSolver:
    """The solver object with ``update`` replaced; everything else (views, params, dynamics) is the solver's own."""

    def __init__(self, opt, update):
        object.__setattr__(self, "_opt", opt)
        object.__setattr__(self, "update", update)

    def __getattr__(self, name):
        return getattr(self._opt, name)

    def __setattr__(self, name, value):
        setattr(self._opt, name, value)
