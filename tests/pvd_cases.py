"""Synthetic planning-cycle inputs for the path/velocity-decomposition glue: the slice of tpl's
environment objects PathSmoothing / PathOptim / VelocityOptim read, as plain namespaces.  Shared by the fixture
generator (tests/golden/make_pvd_goldens.py, which feeds them to the REFERENCE classes) and by the tests (which feed
the same inputs to tpl_amd.planning)."""
from types import SimpleNamespace as NS

import numpy as np

N_REF = 260
STEP = 0.5


def ref_path(n=N_REF, step=STEP, v_max=9.0):
    """(x, y, orientation, s, curvature, v_max): an S-shaped lane, speed limit dipping around s = 70 m."""
    s = np.arange(n) * step
    k = 0.012 * np.sin(s / 18.0)
    phi = np.concatenate([[0.0], np.cumsum(k[:-1] * step)]) + 0.1
    x = 5.0 + np.concatenate([[0.0], np.cumsum(np.cos(phi[:-1]) * step)])
    y = -3.0 + np.concatenate([[0.0], np.cumsum(np.sin(phi[:-1]) * step)])
    v = v_max - 3.0 * np.exp(-((s - 70.0) / 12.0) ** 2)
    return np.stack([x, y, phi, s, k, v], axis=1)


def frenet_point(path, s, d):
    i = min(int(s / STEP), len(path) - 1)
    return np.array([path[i, 0] - np.sin(path[i, 2]) * d, path[i, 1] + np.cos(path[i, 2]) * d])


def hull_at(path, s, d, length=4.5, width=1.8):
    return np.array([frenet_point(path, s + a, d + b) for a, b in
                     ((-length / 2, -width / 2), (length / 2, -width / 2), (length / 2, width / 2), (-length / 2, width / 2))])


def vehicle(path, s=2.0, d=0.3, v=7.0, a=0.2):
    p = frenet_point(path, s, d)
    return NS(x=float(p[0]), y=float(p[1]), v=v, a=a, width=1.9, rear_axis_to_front=3.9, rear_axis_to_rear=1.0)


def environment(cycle=0, with_objects=True):
    """Planning cycle `cycle` of a short drive: the local map window moves 2 reference steps per cycle."""
    full = ref_path(N_REF + 40)
    shift = 2 * cycle
    path = full[shift:shift + N_REF].copy()
    path[:, 3] -= path[0, 3]
    n = len(path)
    d_left = 3.2 + 0.4 * np.sin(np.arange(n) / 25.0)           # lane bounds, positive to the left / right
    d_right = 2.9 + 0.3 * np.cos(np.arange(n) / 31.0)
    local_map = NS(path=path, steps_ref=n, step_size_ref=STEP, d_left=d_left, d_right=d_right,
                   shift_idx_start_ref=(-1 if cycle == 0 else 2))
    veh = vehicle(path, s=2.0, d=0.3 - 0.05 * cycle, v=7.0 + 0.1 * cycle, a=0.2)
    predicted, tracks = [], []
    if with_objects:
        # a slow car half in the lane on the right, 40 m ahead: to be passed on the left
        s_obj = 40.0 - 2 * cycle * STEP + 1.5 * cycle * 0.25
        hull = hull_at(path, s_obj, -1.9)
        predicted.append(NS(hull=hull, v=1.5, evade="", predictions=[NS(uuid_assoc_map="map-a", cos_angle_dist=0.98)]))
        # a leader on the lane centre, 75 m ahead, 5 m/s
        s_lead = 75.0 + 1.2 * cycle
        pos = frenet_point(path, s_lead, 0.1)
        i = min(int(s_lead / STEP), n - 1)
        tracks.append(NS(pos=pos, hull=hull_at(path, s_lead, 0.1), hull_radius=2.5, v=5.0, yaw=float(path[i, 2])))
    man_time_cons = [(frenet_point(path, 110.0, 0.0), 14.0, 30.0)] if with_objects else []
    return NS(t=10.0 + 0.5 * cycle, reset_counter=1, local_map=local_map, local_map_behind=NS(uuid="map-a"),
              vehicle_state=veh, predicted=predicted, man_time_cons=man_time_cons, man_vel_cons=[],
              get_all_tracks=lambda: tracks)


def shared_environment(cycle=0, with_objects=True):
    """The same cycle as the planner plugin sees it: tpl's shared environment state (lock(), tracks by source)."""
    from contextlib import contextmanager
    env = environment(cycle, with_objects)
    tracks = env.get_all_tracks()
    del env.get_all_tracks
    env.tracks = NS(lidar=tracks)
    env.selected_map = None
    env.local_map.intersection_paths = []

    @contextmanager
    def lock():
        yield
    env.lock = lock
    return env


def default_params(kind):
    """The `path_vel_decomp_planner.params` blocks of data/params/planning/default/state.json (numbers committed
    here as data), as attribute objects."""
    if kind == "path_smoothing":
        return NS(horizon=250, step=0.5, ref_step=0.5, k_min=-1.0, k_max=1.0, w_pos=1.0, w_k=0.1)
    if kind == "path_optim":
        return NS(horizon=250, step=0.5, ref_step=0.5, min_d_dd=-2.5, max_d_dd=2.5, max_lat_acc=2.0, lateral_min_gap=1.0,
                  offset_center_line=0.0, fix_seconds=1.0, evasion_lon_d_safe=1, evasion_lat_d_safe=0.5,
                  evasion_lat_d_comf=2.0, evasion_sharpness=20.0, cost_func=NS(w_d=0.01, w_v_d=1.0, w_a_d=200.0, w_k=1.0))
    if kind == "velocity_optim":
        return NS(horizon=250, step=0.5, ref_step=0.5, a_min=-2.0, a_max=2.0, j_min=-1.5, j_max=1.5, max_a_total=5.0,
                  max_lat_acc=2.0, max_traffic_light_stop_acc=2.0, d_lat_leader_safe=0.2, dt_safe=1.5, min_d_safe=2.0,
                  min_v_profile=1.0, time_constr_alpha=10.0, time_constr_beta=0.005, cost_func=NS(p_v=0.1, p_a=1.0))
    raise KeyError(kind)


def rampify_velocity_cases(seed=0, n=6):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        m = int(rng.integers(20, 260))
        s = np.arange(m) * 0.5
        lim = 10.0 - rng.uniform(2, 9) * np.exp(-((s - rng.uniform(5, 5 + m * 0.2)) / rng.uniform(4, 20)) ** 2)
        if i % 2:
            lim[int(m * 0.7):] = 0.0                               # a stop: clamped to v_min inside
        v0 = None if i == 4 else float(rng.uniform(0.5, 9.0))
        a0 = None if i in (4, 5) else float(rng.uniform(-1.0, 1.0))
        cases.append(dict(v0=v0, a0=a0, lim_v=lim, a_min=-2.0, a_max=2.0, j_min=-1.5, j_max=1.5, v_min=1.0, step=0.5))
    return cases


def rampify_lateral_cases(seed=1, n=5):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        h = int(rng.integers(30, 250))
        path = ref_path(h + int(rng.integers(0, 5)))
        lower = -1.8 + rng.uniform(0.5, 2.5) * np.exp(-((np.arange(h) * 0.5 - rng.uniform(10, 60)) / rng.uniform(3, 10)) ** 2)
        upper = np.full(h, 2.2) - 0.3 * np.sin(np.arange(h) / 15.0)
        cases.append(dict(step=0.5, horizon=h, evasion_sharpness=float(rng.uniform(5, 30)), proj_distance=float(rng.uniform(-0.5, 0.5)),
                          path=path, gap=float(rng.uniform(0.5, 2.0)), lower=lower, upper=upper))
    return cases
