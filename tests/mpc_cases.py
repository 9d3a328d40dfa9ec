"""Closed loops for the trajectory-tracking controller plugin: tplsim's ego model (benchkit.minisim.EgoSim, a restatement
of simulation/core.py:91-135 used on BOTH sides) under a controller object, tracking a fixed planner trajectory along
a curved road.  Shared by tests/golden/make_mpc_goldens.py (reference controller + reference solver) and the tests."""
from contextlib import contextmanager
from types import SimpleNamespace as NS

import numpy as np

from benchkit.minisim import EgoSim

DT = 0.01


def planner_trajectory(v=8.0, n=160, step=0.6, curvature=0.01):
    """What a planner hands the controller: time, s, x, y, orientation, curvature, velocity along an arc that straightens."""
    s = np.arange(n) * step
    k = curvature * np.exp(-s / 60.0)
    phi = 0.03 + np.concatenate([[0.0], np.cumsum(k[:-1] * step)])
    x = np.concatenate([[0.0], np.cumsum(np.cos(phi[:-1]) * step)])
    y = np.concatenate([[0.0], np.cumsum(np.sin(phi[:-1]) * step)])
    vel = v + 1.5 * np.tanh((s - 30.0) / 15.0)
    t = np.concatenate([[0.0], np.cumsum(step / vel[:-1])])
    return NS(time=t, s=s, x=x, y=y, orientation=phi, curvature=k, velocity=vel, acceleration=np.zeros(n), emergency=False)


def planner_trajectory_time(v=8.0, n=120, dt=0.1, curvature=0.008):
    """The same road sampled every `dt` seconds (what the time-indexed controller takes as it is)."""
    vel = v + 1.0 * np.tanh((np.arange(n) * dt - 3.0) / 2.0)
    s = np.concatenate([[0.0], np.cumsum(vel[:-1] * dt)])
    k = curvature * np.exp(-s / 60.0)
    ds = np.diff(s, append=s[-1] + vel[-1] * dt)
    phi = 0.03 + np.concatenate([[0.0], np.cumsum(k[:-1] * ds[:-1])])
    x = np.concatenate([[0.0], np.cumsum(np.cos(phi[:-1]) * ds[:-1])])
    y = np.concatenate([[0.0], np.cumsum(np.sin(phi[:-1]) * ds[:-1])])
    return NS(time=np.arange(n) * dt, s=s, x=x, y=y, orientation=phi, curvature=k, velocity=vel, acceleration=np.zeros(n),
              emergency=False)


@contextmanager
def nolock():
    yield


def vehicle_of(ego, pitch=0.02):
    return NS(x=ego.x, y=ego.y, phi=ego.yaw, delta=ego.steer_angle, v=ego.v, a=ego.a, pitch=pitch,
              wheel_base=ego.wheel_base, delta_max=ego.max_steer_angle, dead_time_steer=ego.steer_dead_time)


CASES = {
    # name: (ego start, parameter overrides, ticks)
    "default": (dict(x=0.0, y=0.5, yaw=0.08, v=6.0), {}, 120),
    "short_horizon_no_dead_time": (dict(x=0.5, y=-0.4, yaw=-0.05, v=9.0, steer_dead_time=0.0, acc_dead_time=0.0),
                                   dict(horizon=20, max_iterations=8), 120),
    # the stopped vehicle makes the tracking problem degenerate (v = 0): the loop ends before rounding decides line searches
    "idle_compensation": (dict(x=0.0, y=0.0, yaw=0.03, v=0.2), dict(horizon=30, idle=True), 36),
}


def rebased(plan, t, n=80, dt=0.1):
    """The plan as a planner would hand it over at time t: sampled every `dt` from NOW on (the time-indexed controller
    reads sample k as "where to be k * dt from now")."""
    ts = t + np.arange(n) * dt
    f = lambda a: np.interp(ts, plan.time, a)
    return NS(time=ts, s=f(plan.s), x=f(plan.x), y=f(plan.y), orientation=f(plan.orientation), curvature=f(plan.curvature),
              velocity=f(plan.velocity), acceleration=np.zeros(n), emergency=False)


TIME_CASES = {
    "time_default": (dict(x=-1.58, y=0.3, yaw=0.05, v=7.1), {}, 100),
    "time_no_dead_time": (dict(x=-0.95, y=-0.2, yaw=0.03, v=7.6, steer_dead_time=0.0, acc_dead_time=0.0),
                          dict(horizon=25, max_iterations=10, cog_pos=0.3), 100),
}


def run(controller_factory, name, on_tick=None):
    """-> controls float64[ticks, 2], ego trace float64[ticks, 4], the controller."""
    ego_kw, over, ticks = CASES[name] if name in CASES else TIME_CASES[name]
    ego = EgoSim(**ego_kw)
    shared = NS()
    con = controller_factory(shared, nolock)
    p = shared.params
    for k, v in over.items():
        if k == "idle":
            p.idle_comp.active = True
        else:
            setattr(p, k, v)
    traj = planner_trajectory_time() if name in TIME_CASES else planner_trajectory(v=0.05 if over.get("idle") else 8.0)
    controls, trace, extra = np.zeros((ticks, 2)), np.zeros((ticks, 4)), []
    t = 0.0
    for k in range(ticks):
        now = rebased(traj, t) if name in TIME_CASES else traj
        (acc, steer), con_traj = con.update(NS(t=t, vehicle=vehicle_of(ego), trajectory=now))
        ego.control_acc, ego.control_steer = acc, steer
        for _ in range(2):
            ego.step(t, DT / 2)
            t += DT / 2
        controls[k] = (acc, steer)
        trace[k] = (ego.x, ego.y, ego.yaw, ego.v)
        if k in (0, ticks // 2, ticks - 1):
            cols = [con_traj.time, con_traj.x, con_traj.y, con_traj.orientation, con_traj.velocity, con_traj.curvature]
            if name in CASES:
                cols += [con_traj.s, con_traj.acceleration]
            extra.append(np.stack(cols, axis=1))
        if on_tick:
            on_tick(k, con)
    return controls, trace, extra, con
