"""Synthetic drives for the planner plugins (DpEnv, PolyLatDpLonPlanner, DpLatLonPlanner): the slice of tpl's shared
environment state a planner reads, as plain namespaces, cycle by cycle.  Used by the fixture generator
(tests/golden/make_dp_planner_goldens.py feeds the cycles to the REFERENCE classes), by the tests (the same cycles to
tpl_amd.planning.dyn_prog), by the planner-cycle latency leg of bench.py and by examples/closed_loop.py.  Pure numpy."""
from contextlib import contextmanager
from types import SimpleNamespace as NS

import numpy as np

STEP = 0.5
N_REF = 460                 # 230 m of local map: longer than the environment's s_max
DT = 0.1                    # planner cycle


def road(n, step=STEP):
    """(x, y, orientation, s, curvature, v_max) of a gentle S-bend, 13.9 m/s limit with a 9 m/s stretch."""
    s = np.arange(n) * step
    k = 0.006 * np.sin(s / 35.0)
    phi = 0.2 + np.concatenate([[0.0], np.cumsum(k[:-1] * step)])
    x = 20.0 + np.concatenate([[0.0], np.cumsum(np.cos(phi[:-1]) * step)])
    y = -8.0 + np.concatenate([[0.0], np.cumsum(np.sin(phi[:-1]) * step)])
    v = np.where((s > 150.0) & (s < 190.0), 9.0, 13.9)
    return np.stack([x, y, phi, s, k, v], axis=1)


FULL = road(N_REF + 400)


def frenet(path, s, d):
    i = int(np.clip(s / STEP, 0, len(path) - 2))
    a = s / STEP - i
    p = path[i, :2] * (1 - a) + path[i + 1, :2] * a
    h = path[i, 2]
    return np.array([p[0] - np.sin(h) * d, p[1] + np.cos(h) * d]), h


def box(path, s, d, length=4.5, width=1.8):
    return np.array([frenet(path, s + a, d + b)[0] for a, b in
                     ((-length / 2, -width / 2), (length / 2, -width / 2), (length / 2, width / 2), (-length / 2, width / 2))])


def prediction(path, s0, d, v, horizon=10.0, dt=1.0):
    ts = np.arange(0.0, horizon + 1e-9, dt)
    st = np.zeros((len(ts), 5))
    st[:, 0] = ts
    for i, t in enumerate(ts):
        p, h = frenet(path, s0 + v * t, d)
        st[i, 1:3] = p
        st[i, 3] = h
    st[:, 4] = v
    return st


class Drive:
    """A vehicle following the road at `v_ego`; the local map window starts `behind` metres behind it and moves in
    whole reference steps.  Objects: a slow car ahead on the right edge, a leader on the lane centre, a parked car."""

    def __init__(self, v_ego=9.0, automated=True, crossing=False, reset_at=None, jump_at=None, blocked=False):
        """crossing: an intersecting road 70 m ahead with two cars on it (one accelerating) and its stop line on the
        map; reset_at: cycle from which the environment's reset counter is one higher; jump_at: cycle at which the
        vehicle is found 3 m beside where it should be (a localisation jump)."""
        self.v_ego = v_ego
        self.automated = automated
        self.crossing, self.reset_at, self.jump_at = crossing, reset_at, jump_at
        self.blocked = blocked              # a standing obstacle across the whole lane 22 m ahead: no valid plan exists
        self.s_ego = 30.0
        self.idx_prev = None
        self.t = 100.0

    @contextmanager
    def _nolock(self):
        yield

    def cycle(self, k, follow=None, man_time=False):
        """Shared environment state of cycle k.  `follow`: trajectory of the previous cycle to place the vehicle on
        (closed loop); None keeps the nominal constant-speed drive."""
        t = self.t + k * DT
        s_ego = self.s_ego + self.v_ego * k * DT
        idx0 = int((s_ego - 10.0) / STEP)
        shift = 0 if self.idx_prev is None else idx0 - self.idx_prev
        self.idx_prev = idx0
        path = FULL[idx0:idx0 + N_REF].copy()
        path[:, 3] -= path[0, 3]
        n = len(path)
        d_left = np.full(n, 3.5) + 0.3 * np.sin(np.arange(n) / 40.0)
        d_right = np.full(n, 3.0)
        s_loc = s_ego - idx0 * STEP
        pos, h = frenet(path, s_loc, 0.2)
        veh = NS(x=float(pos[0]) - np.cos(h) * 1.58, y=float(pos[1]) - np.sin(h) * 1.58, phi=float(h), v=self.v_ego, a=0.1,
                 width=1.9, wheel_base=3.16, rear_axis_to_front=3.9, rear_axis_to_rear=1.0, automated=self.automated)
        if follow is not None:
            veh.x, veh.y, veh.phi, veh.v, veh.a = follow
        if self.jump_at is not None and k == self.jump_at:
            veh.x -= np.sin(h) * 3.0
            veh.y += np.cos(h) * 3.0
        local_map = NS(uuid="map-local", name="local_map", path=path, steps_ref=n, step_size_ref=STEP, d_left=d_left,
                       d_right=d_right, shift_idx_start_ref=shift, intersection_paths=[])
        behind = NS(uuid="map-behind", name="local_map_behind", path=path.copy())
        objs = []
        if self.blocked:
            wall = box(path, s_loc + 22.0, 0.0, length=3.0, width=9.0)
            objs.append(NS(hull=wall, stationary=True, a=0.0, v=0.0,
                           predictions=[NS(uuid_assoc_map="map-behind", states=prediction(path, s_loc + 22.0, 0.0, 0.0))]))
        if self.crossing:
            # a road crossing ours at right angles 100 m (absolute) along it; its stop line 8 m before the crossing
            s_x = 100.0 - idx0 * STEP
            if 12.0 < s_x < n * STEP - 30.0:
                centre, hx = frenet(path, s_x, 0.0)
                along = np.array([-np.sin(hx), np.cos(hx)])
                d = np.arange(-60.0, 60.0, STEP)
                cross = np.zeros((len(d), 6))
                cross[:, :2] = centre + d[:, None] * along
                cross[:, 2], cross[:, 3], cross[:, 5] = hx + np.pi / 2, d + 60.0, 13.9
                seg = NS(uuid="map-cross", name="crossing", path=cross)
                stop_pos = frenet(path, s_x - 8.0, 0.0)[0]
                from tpl_amd.planning.geometry import project
                local_map.intersection_paths = [NS(stop_pos=stop_pos, stop_proj=project(path[:, :2], stop_pos), map_segment=seg)]
                for (d0, v, acc) in ((-40.0 + 0.0, 6.0, 0.0), (-25.0, 3.0, 1.5)):
                    sc = 60.0 + d0 + v * k * DT
                    objs.append(NS(hull=box(cross, sc, 0.0), stationary=False, a=acc, v=v,
                                   predictions=[NS(uuid_assoc_map="map-cross", states=prediction(cross, sc, 0.0, v))]))
        for (s_abs0, d, v, stationary, acc) in ((95.0, -2.2, 2.0, False, 0.0), (130.0, 0.1, 6.0, False, None),
                                               (70.0, 2.9, 0.0, True, 0.0)):
            s_obj = s_abs0 + v * k * DT - idx0 * STEP
            if not 5.0 < s_obj < n * STEP - 20.0:
                continue
            objs.append(NS(hull=box(path, s_obj, d), stationary=stationary, a=acc, v=v,
                           predictions=[NS(uuid_assoc_map="map-behind", states=prediction(path, s_obj, d, v)),
                                        NS(uuid_assoc_map="map-unknown", states=prediction(path, s_obj, d, v))]))
        sh = NS(t=t, reset_counter=2 if (self.reset_at is not None and k >= self.reset_at) else 1, vehicle_state=veh, local_map=local_map, local_map_behind=behind, selected_map=None,
                tracks=[], predicted=objs,
                man_time_cons=[(frenet(path, s_loc + 60.0, 0.0)[0], t + 6.0, t + 30.0)] if man_time else [])
        sh.lock = self._nolock
        return sh


def trajectory_row(traj):
    """What a cycle's result is compared on."""
    if traj.emergency:
        return None
    return np.stack([traj.time, traj.s, traj.x, traj.y, traj.velocity, traj.acceleration, traj.orientation, traj.curvature], axis=1)
