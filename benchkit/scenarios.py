"""Synthetic inputs for the DP planner ("Synthetic inputs -- DP").

Pure input generation (numpy); nothing here computes a plan.  The numbers come
from the reference's default parameter file
(data/params/planning/default/state.json -> poly_lat_dp_lon_planner.params) and
its straight-road test scenarios (data/scenarios/test/overtake_*): committed
here as plain constants.
"""

import numpy as np

from tpl_amd.dyn_prog import DynProgEnvParams, DynProgLonPlannerParams, LonState

# (s0 [m], v [m/s], length [m]) of the cars ahead of the ego vehicle in
# data/scenarios/test/overtake_multiple_dynamic_obstacles/state.json
# (straight road, ego at s=0 with 13.0 m/s; Euclidean distance = station offset)
OBSTACLES_OVERTAKE = ((96.7, 3.0, 4.0), (113.9, 3.0, 4.0), (128.5, 3.0, 4.0), (198.6, 3.0, 4.0))
# ..._with_traffic adds one faster car
OBSTACLES_OVERTAKE_TRAFFIC = OBSTACLES_OVERTAKE + ((220.4, 10.0, 4.0),)

GRIDS = {
    "default": (201, 37, 20, 10),        # reference default grid, t_steps
    "config2": (128, 64, 32, 50),        # BASELINE configs[1]
    "target": (256, 128, 64, 50),        # grid of the >=10k sweeps/s target
    "config4": (512, 256, 128, 100),     # BASELINE configs[3]
}


def lon_params(s_steps=201, v_steps=37, a_steps=20, t_steps=10, dt=1.0, dt_start=None):
    """Planner parameters with the reference's default ranges and weights."""
    p = DynProgLonPlannerParams()
    p.s_min, p.s_max = 0.0, 200.0
    p.v_min, p.v_max = 0.0, 36.0
    p.a_min, p.a_max = -3.0, 1.5
    p.j_min, p.j_max = -3.0, 2.0
    p.t_steps, p.s_steps, p.v_steps, p.a_steps = t_steps, s_steps, v_steps, a_steps
    p.dt = dt
    p.dt_start = dt if dt_start is None else dt_start
    p.time_gap, p.gap_min = 1.0, 1.0
    p.w_progress, p.w_a, p.w_j, p.w_snap, p.w_safety_dist = 0.1, 1.0, 1.0, 0.5, 10.0
    p.path_step_size, p.path_steps = 1.0, 200
    p.width_veh, p.length_veh = 2.0, 4.895
    return p


def env_params(t_steps=10, s_steps=201, l_steps=31, dt=1.0, dt_start=None):
    e = DynProgEnvParams()
    e.t_min, e.t_max = 0.0, float(t_steps) * dt
    e.l_min, e.l_max = -5.0, 5.0
    e.s_min, e.s_max = 0.0, 200.0
    e.dt = dt
    e.dt_start = dt if dt_start is None else dt_start
    e.t_steps, e.l_steps, e.s_steps = t_steps, l_steps, s_steps
    e.dilation = 1.4142135623730951
    return e


def straight_path(path_steps=200, step=1.0, l=0.0):
    """float32[path_steps, 7] = PathState{x, y, s, l, k, v_max, distance}: straight
    road, s = distance, speed limit dipping to 8.9 m/s around d = 120 m."""
    d = np.arange(path_steps, dtype=np.float64) * step
    path = np.zeros((path_steps, 7), dtype=np.float64)
    path[:, 0] = d
    path[:, 2] = d
    path[:, 3] = l
    path[:, 5] = 13.9 - 5.0 * np.exp(-((d - 120.0) / 15.0) ** 2)
    path[:, 6] = d
    return path.astype(np.float32)


def occ_map(ep, obstacles=(), sweep_length=0.5, half_width=1.5):
    """float32[l_steps, s_steps, t_steps] Frenet occupancy (0 free, 1 occupied).

    Obstacle k occupies, in time slice i (slice 0 <-> t = 0, slice i >= 1 <->
    t = dt_start + (i-1) dt, the inverse of env.cu:258), the stations
    [s0 + v t - len/2, s0 + v t + len/2 + max(sweep_length, dt) v] and the
    lateral cells with |l| <= half_width.
    """
    occ = np.zeros((ep.l_steps, ep.s_steps, ep.t_steps), dtype=np.float32)
    s = ep.s_min + np.arange(ep.s_steps) * ((ep.s_max - ep.s_min) / (ep.s_steps - 1))
    l = ep.l_min + np.arange(ep.l_steps) * ((ep.l_max - ep.l_min) / (ep.l_steps - 1))
    lmask = np.abs(l) <= half_width
    for i in range(ep.t_steps):
        t = 0.0 if i == 0 else ep.dt_start + (i - 1) * ep.dt
        for (s0, v, length) in obstacles:
            lo = s0 + v * t - 0.5 * length
            hi = s0 + v * t + 0.5 * length + max(sweep_length, ep.dt) * v
            smask = (s >= lo) & (s <= hi)
            occ[np.ix_(lmask, smask, [i])] = 1.0
    return occ


def random_obstacles(seed=4, k=8):
    """Seeded stress variant: k cars, s0 ~ U(30,180), v ~ U(0,10)."""
    rng = np.random.default_rng(seed)
    s0 = rng.uniform(30.0, 180.0, size=k)
    v = rng.uniform(0.0, 10.0, size=k)
    return tuple((float(a), float(b), 4.0) for a, b in zip(s0, v))


def initial_state(s=0.0, v=8.0, a=0.0):
    st = LonState()
    st.s, st.v, st.a = s, v, a
    return st


def random_initial_states(n, seed=5):
    """Batch variant: v0 ~ U(2,14), a0 ~ U(-1,1)."""
    rng = np.random.default_rng(seed)
    v0 = rng.uniform(2.0, 14.0, size=n)
    a0 = rng.uniform(-1.0, 1.0, size=n)
    return [initial_state(0.0, float(v), float(a)) for v, a in zip(v0, a0)]


# --------------------------------------------------------------------------
# inputs of the occupancy pipeline: reference lines and the
# triangle soups tpl's prediction code hands to insert_geometry
# --------------------------------------------------------------------------

def ref_line_straight(n=440, step=0.5, x0=100.0, y0=-40.0, heading=0.3, d_left=4.5, d_right=4.5,
                      v_max=13.9, semantic_range=None):
    """float64[n, 9] = RefLinePoint{x, y, heading, s, k, v_max, d_left, d_right, semantic}:
    a straight road through (x0, y0)."""
    s = np.arange(n, dtype=np.float64) * step
    rl = np.zeros((n, 9))
    rl[:, 0] = x0 + s * np.cos(heading)
    rl[:, 1] = y0 + s * np.sin(heading)
    rl[:, 2] = heading
    rl[:, 3] = s
    rl[:, 5] = v_max
    rl[:, 6] = d_left
    rl[:, 7] = d_right
    if semantic_range is not None:
        rl[semantic_range[0]:semantic_range[1], 8] = 1.0
    return rl


def ref_line_arc(n=440, step=0.5, radius=120.0, x0=-30.0, y0=10.0, heading0=2.9, d_left=5.0, d_right=3.0,
                 v_max=11.0, semantic_range=None):
    """A left-hand arc whose heading crosses +-pi (exercises the short-angle interpolation)."""
    s = np.arange(n, dtype=np.float64) * step
    h = heading0 + s / radius
    rl = np.zeros((n, 9))
    rl[:, 0] = x0 + radius * (np.sin(h) - np.sin(heading0))
    rl[:, 1] = y0 - radius * (np.cos(h) - np.cos(heading0))
    rl[:, 2] = np.arctan2(np.sin(h), np.cos(h))
    rl[:, 3] = s
    rl[:, 4] = 1.0 / radius
    rl[:, 5] = v_max
    rl[:, 6] = d_left
    rl[:, 7] = d_right
    if semantic_range is not None:
        rl[semantic_range[0]:semantic_range[1], 8] = 1.0
    return rl


def frenet_to_cart(rl, step, s, l):
    """Point at station s, lateral offset l of a reference line (linear between samples)."""
    a = np.clip(s / step, 0, len(rl) - 1)
    i0 = int(np.floor(a))
    i1 = min(i0 + 1, len(rl) - 1)
    w = a - i0
    x = rl[i0, 0] * (1 - w) + rl[i1, 0] * w
    y = rl[i0, 1] * (1 - w) + rl[i1, 1] * w
    h = rl[i0, 2]
    return x - l * np.sin(h), y + l * np.cos(h)


def prediction_soup(rl, step, s0, v, length, l_lo, l_hi, ts, sweep=0.5, station_step=5.0):
    """Triangle soup float64[n, 3] = (x, y, t) shaped like tpl's prediction geometry
    (utils.cpp:576-693): per time ts[i] the object covers stations
    [s0 + v t - length/2, s0 + v t + length/2 + max(sweep, dt) v], cut into quads of at most
    ``station_step`` metres, two triangles each, between the lateral offsets l_lo and l_hi."""
    out = []
    for i, t in enumerate(ts):
        dt = (ts[i + 1] - t) if i + 1 < len(ts) else (ts[i] - ts[i - 1] if i > 0 else 1.0)
        a = s0 + v * t - 0.5 * length
        b = s0 + v * t + 0.5 * length + max(sweep, dt) * v
        k = max(1, int(np.ceil((b - a) / station_step)))
        cuts = np.linspace(a, b, k + 1)
        for c0, c1 in zip(cuts[:-1], cuts[1:]):
            p00 = frenet_to_cart(rl, step, c0, l_lo)
            p10 = frenet_to_cart(rl, step, c1, l_lo)
            p01 = frenet_to_cart(rl, step, c0, l_hi)
            p11 = frenet_to_cart(rl, step, c1, l_hi)
            for p in (p00, p10, p01, p01, p10, p11):
                out.append((p[0], p[1], t))
    return np.array(out, dtype=np.float64).reshape(-1, 3)


def env_scene(kind="straight", seed=0, n_objects=6, t_steps=10, dt=1.0):
    """(ref_line, step, [(geom, stationary), ...]) for the occupancy pipeline tests and bench."""
    rng = np.random.default_rng(seed)
    step = 0.5
    if kind == "straight":
        rl = ref_line_straight(step=step, semantic_range=(150, 190))
    else:
        rl = ref_line_arc(step=step, semantic_range=(120, 170))
    ts = np.array([0.0] + [dt + i * dt for i in range(t_steps - 1)])
    objs = []
    for k in range(n_objects):
        stationary = (k % 3 == 2)
        s0 = float(rng.uniform(20.0, 180.0))
        v = 0.0 if stationary else float(rng.uniform(0.5, 9.0))
        lc = float(rng.uniform(-3.0, 3.0))
        objs.append((prediction_soup(rl, step, s0, v, 4.5, lc - 1.0, lc + 1.0, ts), stationary))
    return rl, step, objs
