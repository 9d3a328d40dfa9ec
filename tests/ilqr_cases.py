"""Problem instances for the iLQR parity tests.

Each case configures an ``Optim``-like object through the attribute surface
the reference's generated solver type exposes (optim.c:1654-1677, 1824-1882),
so the SAME function drives the reference solver (when the goldens are made,
tests/golden/make_ilqr_goldens.py), the CPU oracle and the HIP solver.

The settings are the ones tpl's callers use (cited per case).
"""
import numpy as np

# data/params/control/default/state.json -> model_predictive_controller.params.cost_function
MPC_COSTS = dict(pd=10.0, pv=5.0, pdelta=0.0, min_pdelta_dot=0.1, pdelta_dot=0.0, min_p_phi_dot=0.0,
                 p_phi_dot=0.05, p_phi=1000.0, p_phi_ref_dot_diff=0.01, pa=2.0, pj=0.5)
# ... -> model_predictive_controller_time.params.cost_function
MPC_TIME_COSTS = dict(pd=5.0, pv=5.0, pdelta=0.0, min_pdelta_dot=0.1, pdelta_dot=0.1, min_p_phi_dot=0.0,
                      p_phi_dot=0.0, p_phi=0.0, pa=2.0, pj=0.5)


def _set_params(opt, **kw):
    for k, v in kw.items():
        setattr(opt.params, k, v)


def _arc(n, step, curvature, v):
    """Reference course: n points every `step` metres on an arc of given curvature."""
    s = np.arange(n) * step
    if abs(curvature) < 1e-12:
        x, y, phi = s.copy(), np.zeros(n), np.zeros(n)
    else:
        phi = s * curvature
        x = np.sin(phi) / curvature
        y = (1.0 - np.cos(phi)) / curvature
    return x, y, phi, np.full(n, curvature), np.full(n, float(v))


def mpc(opt, seed=0, curvature=0.0, horizon=50, min_rel=1e-6, v_ref=8.0):
    """trajectory_tracking_mpc as ModelPredictiveController sets it up
    (control/model_predictive_controller.py:69-76, 134-155, 192); sizes of
    SURVEY.md 8(d)."""
    rng = np.random.default_rng(seed)
    opt.integrator_type = opt.HEUN
    opt.lg_mult_limit = 0.0
    opt.horizon = horizon
    opt.step = 0.05
    opt.max_iterations = 20
    opt.max_lg_iterations = 1
    opt.min_rel_cost_change = min_rel
    opt.barrier_weight[:] = 10000.0
    opt.lagrange_multiplier[:] = 0.0
    opt.u_min[..., 0] = -3.0             # "..." so the same lines also configure a batched solver
    opt.u_max[..., 0] = 1.5
    opt.u_min[..., 1] = -1.0
    opt.u_max[..., 1] = 1.0
    x, y, phi, k, v = _arc(100, 0.5, curvature, v_ref)
    _set_params(opt, v_ch=32.0, max_delta=0.7, min_acc=-3.0, max_acc=2.0, l=3.165, ref_step=0.5,
                a_offset=0.0, ref_x=x, ref_y=y, ref_phi=phi, ref_k=k, ref_v=v, **MPC_COSTS)
    x0 = np.array([0.0, rng.uniform(-1, 1), rng.uniform(-0.2, 0.2), 0.0, rng.uniform(4, 10), 0.0, 0.0])
    opt.x[..., 0, :] = x0
    opt.u[:] = 0.0


def mpc_time(opt, seed=0):
    """trajectory_tracking_mpc_time (control/model_predictive_controller_time.py:73-79, 120-142)."""
    rng = np.random.default_rng(100 + seed)
    opt.integrator_type = opt.HEUN
    opt.lg_mult_limit = 0.0
    opt.horizon = 40
    opt.step = 0.05
    opt.max_iterations = 20
    opt.barrier_weight[:] = 10000.0
    opt.lagrange_multiplier[:] = 0.0
    opt.u_min[:, 0] = -3.0
    opt.u_max[:, 0] = 1.5
    opt.u_min[:, 1] = -1.0
    opt.u_max[:, 1] = 1.0
    n, ref_dt, v = 60, 0.1, 7.0
    t = np.arange(n) * ref_dt
    curv = 0.03
    phi = v * t * curv
    _set_params(opt, v_ch=32.0, max_delta=0.7, min_acc=-3.0, max_acc=3.0, l=3.165, cog_pos=0.5,
                ref_dt=ref_dt, ref_t_offset=0.03, a_offset=0.1,
                ref_x=np.sin(phi) / curv, ref_y=(1 - np.cos(phi)) / curv, ref_phi=phi, ref_v=np.full(n, v),
                **MPC_TIME_COSTS)
    opt.x[0] = np.array([0.0, rng.uniform(-0.5, 0.5), rng.uniform(-0.1, 0.1), 0.0, rng.uniform(5, 9), 0.0])
    opt.u[:] = 0.0


def lateral(opt, seed=0, horizon=120):
    """lateral_profile as PathOptim runs it (planning/path_vel_decomp/path_optim.py:101-104,
    125-133, 292-299): corridor with an obstacle bump on the right."""
    rng = np.random.default_rng(200 + seed)
    opt.integrator_type = opt.EULER
    opt.lg_mult_limit = 0.0
    opt.barrier_weight[:] = 1000.0
    opt.horizon = horizon
    opt.step = 0.5
    opt.max_iterations = 5
    opt.lagrange_multiplier[:] = 0.0
    opt.u_min[:] = -0.5
    opt.u_max[:] = 0.5
    s = np.arange(horizon) * 0.5
    lower = -2.0 + 2.6 * np.exp(-((s - 30.0) / 6.0) ** 2)       # obstacle pushes the lower bound up
    upper = np.full(horizon, 2.5)
    _set_params(opt, ref_step=0.5, w_d=1.0, w_v_d=10.0, w_a_d=100.0, w_k=50.0,
                k_ref=0.01 * np.sin(s / 20.0), d_offset=np.zeros(horizon),
                d_lower_constr=lower, d_upper_constr=upper)
    opt.x[0, 0] = rng.uniform(-0.5, 0.5)
    opt.x[0, 1] = 0.0
    opt.u[:] = 0.0


def velocity_space(opt, seed=0, horizon=100):
    """velocity_profile_space as VelocityOptim runs it
    (planning/path_vel_decomp/velocity_optim.py:62-66, 153-162, 232-240, 290)."""
    rng = np.random.default_rng(300 + seed)
    opt.integrator_type = opt.EULER
    opt.max_iterations = 20
    opt.lg_mult_limit = 0.1
    opt.barrier_weight[:] = 1000.0
    opt.lagrange_multiplier[:] = 0.0
    opt.horizon = horizon
    opt.step = 0.5
    opt.u_max[:] = 1.5
    opt.u_min[:] = -3.0
    s = np.arange(horizon) * 0.5
    t_min = np.zeros(horizon)
    t_min[60] = 9.0                                               # "not before t = 9 s at s = 30 m"
    _set_params(opt, ref_step=0.5, max_a_total=3.0, p_v=1.0, p_a=2.0,
                ref_v=10.0 - 4.0 * np.exp(-((s - 25.0) / 8.0) ** 2), ref_k=0.02 * np.sin(s / 15.0),
                ref_t_max=np.ones(horizon) * 10e10, ref_t_min=t_min, ref_t_offset=np.ones(horizon),
                ref_v_weight=np.ones(horizon))
    opt.x[0, 0] = rng.uniform(4.0, 8.0)
    opt.x[0, 1] = 0.0
    opt.u[:] = 0.0


def smoother(opt, seed=0, horizon=80, with_dk=False):
    """ref_line_smoother_k / _dk (planning/path_vel_decomp/path_smoothing.py:26-47)."""
    rng = np.random.default_rng(400 + seed)
    opt.integrator_type = opt.EULER
    opt.lg_mult_limit = 0.1
    opt.barrier_weight[:] = 1000.0
    opt.horizon = horizon
    opt.step = 0.5
    opt.max_iterations = 5
    opt.u_min[:] = -0.2
    opt.u_max[:] = 0.2
    s = np.arange(horizon + 10) * 0.5
    phi = 0.3 * np.sin(s / 12.0)
    x = np.cumsum(np.cos(phi)) * 0.5 + rng.normal(0, 0.05, s.shape)
    y = np.cumsum(np.sin(phi)) * 0.5 + rng.normal(0, 0.05, s.shape)
    if with_dk:
        _set_params(opt, w_pos=1.0, w_k=10.0, w_dk=100.0, s_start=0.25, ref_x=x, ref_y=y, ref_step=0.5)
        opt.x[0] = np.array([x[0], y[0], phi[0], 0.0])
    else:
        _set_params(opt, w_pos=1.0, w_k=10.0, ref_x=x, ref_y=y, ref_step=0.5)
        opt.x[0] = np.array([x[0], y[0], phi[0]])
    opt.u[:] = 0.0


def velocity_time(opt, seed=0, horizon=30):
    """velocity_profile_time (optimizers.py:297-349): get_array_value(t) + Piecewise constraints."""
    rng = np.random.default_rng(500 + seed)
    opt.integrator_type = opt.EULER
    opt.lg_mult_limit = 10.0
    opt.barrier_weight[:] = 100.0
    opt.lagrange_multiplier[:] = 0.0
    opt.max_lg_iterations = 2
    opt.horizon = horizon
    opt.step = 0.2
    opt.max_iterations = 8
    opt.u_min[:] = -3.0
    opt.u_max[:] = 2.0
    s_max = np.zeros(horizon + 1)
    s_max[15:] = 40.0
    s_min = np.zeros(horizon + 1)
    s_min[25:] = 20.0
    _set_params(opt, w_v=1.0, w_a=1.0, ref_v=np.linspace(12.0, 6.0, 120), ref_step=0.5,
                ref_s_max=s_max, ref_s_min=s_min)
    opt.x[0] = np.array([0.0, rng.uniform(5.0, 9.0)])
    opt.u[:] = 0.0


def simple_vehicle(opt, seed=0, first_order=False):
    """examples/simple_vehicle/main.py:63-100 (RK4 to cover that integrator)."""
    rng = np.random.default_rng(600 + seed)
    opt.integrator_type = opt.RK4
    opt.step = 0.05
    opt.horizon = 40
    opt.max_iterations = 10
    opt.use_quadratic_terms = not first_order
    opt.u_min[:, 0] = -1.0
    opt.u_max[:, 0] = 5.0
    opt.u_min[:, 1] = -1.0
    opt.u_max[:, 1] = 1.0
    _set_params(opt, x_trg=rng.uniform(3, 6), y_trg=rng.uniform(-2, 2), w_e_pos=1.0, w_e_v=0.01, w_steer=0.1)
    opt.x[0] = np.zeros(3)
    opt.u[:, 0] = 1.0
    opt.u[:, 1] = 0.0


def builtin_probe(opt, seed=0):
    """blerp / lerp_wrap / box_interp / get_array_value (see optimizers.config_builtin_probe)."""
    rng = np.random.default_rng(700 + seed)
    opt.integrator_type = opt.HEUN
    opt.step = 0.1
    opt.horizon = 25
    opt.max_iterations = 6
    opt.lg_mult_limit = 5.0
    opt.barrier_weight[:] = 50.0
    opt.lagrange_multiplier[:] = 0.0
    opt.u_min[:] = -2.0
    opt.u_max[:] = 2.0
    gy, gx = np.meshgrid(np.linspace(-3, 3, 9), np.linspace(-2, 6, 13), indexing="ij")   # [rows = y][cols = x]
    field = 0.3 * (gx - 3.0) ** 2 + 0.5 * (gy - 1.0) ** 2 + 0.2 * np.sin(gx * gy)
    loop_xs = np.arange(8) * 0.5
    _set_params(opt, w_f=1.0, w_u=0.1, x0g=-2.0, y0g=-3.0, dxg=8.0 / 12, dyg=6.0 / 8, loop_len=4.5, loop_dx=0.5,
                field=field, loop_xs=loop_xs, loop_vals=0.2 * np.cos(loop_xs * 1.3),
                steps=np.array([1.0, 0.5, 1.5, 1.0]), gains=np.linspace(1.0, 2.0, 26))
    opt.x[0] = np.array([rng.uniform(-1.0, 0.0), rng.uniform(-1.0, 1.0)])
    opt.u[:] = 0.1


# case name -> (model name, setup function, kwargs, warm-start shift between update 1 and 2)
CASES = {
    "mpc_straight_s0": ("trajectory_tracking_mpc", mpc, dict(seed=0), 1),
    "mpc_straight_s1": ("trajectory_tracking_mpc", mpc, dict(seed=1), 1),
    "mpc_straight_fixed_iters": ("trajectory_tracking_mpc", mpc, dict(seed=2, min_rel=0.0), 0),
    "mpc_curve": ("trajectory_tracking_mpc", mpc, dict(seed=3, curvature=0.02, horizon=60), 2),
    "mpc_time": ("trajectory_tracking_mpc_time", mpc_time, dict(seed=0), 1),
    "lateral": ("lateral_profile", lateral, dict(seed=0), 3),
    "velocity_space": ("velocity_profile_space", velocity_space, dict(seed=0), 2),
    "smoother_k": ("ref_line_smoother_k", smoother, dict(seed=0), 0),
    "smoother_dk": ("ref_line_smoother_dk", smoother, dict(seed=1, with_dk=True), 0),
    "velocity_time": ("velocity_profile_time", velocity_time, dict(seed=0), 1),
    "simple_vehicle": ("simple_vehicle", simple_vehicle, dict(seed=0), 0),
    "simple_vehicle_first_order": ("simple_vehicle", simple_vehicle, dict(seed=1, first_order=True), 0),
    "builtin_probe": ("builtin_probe", builtin_probe, dict(seed=0), 1),
}

ARRAY_FIELDS = ["int_step", "x", "u", "next_int_step", "next_x", "next_u", "prev_x", "prev_u", "prev_k",
                "fx", "fu", "lx", "lu", "lxx", "luu", "lux", "g", "k", "K", "lagrange_multiplier",
                "barrier_weight", "lg_mult_limit", "u_max", "u_min"]
SCALAR_FIELDS = ["traj_costs", "iterations", "lg_iterations", "alpha", "mu", "mu_step",
                 "trajectory_changed", "improved", "termination_condition", "use_quadratic_terms",
                 "dt", "max_iterations", "max_lg_iterations", "opt_start", "min_rel_cost_change",
                 "horizon", "integrator_type"]
INPUT_ARRAYS = ["x", "u", "lagrange_multiplier", "barrier_weight", "lg_mult_limit", "u_max", "u_min"]
INPUT_SCALARS = ["use_quadratic_terms", "dt", "max_iterations", "max_lg_iterations", "opt_start",
                 "min_rel_cost_change", "horizon", "integrator_type", "mu", "mu_step"]


def snapshot(opt, fields=None):
    out = {}
    for f in (fields or ARRAY_FIELDS):
        out[f] = np.array(getattr(opt, f), dtype=np.float64, copy=True)
    for f in SCALAR_FIELDS:
        out[f] = np.float64(getattr(opt, f))
    return out


def param_snapshot(opt):
    out = {}
    for name in opt.params.__slots__:
        out["param." + name] = np.array(getattr(opt.params, name), dtype=np.float64, copy=True)
    return out


def apply_inputs(opt, data):
    """Configure `opt` from the input part of a golden file."""
    # horizon first: it decides the shape of every view
    opt.horizon = int(data["in.horizon"])
    for f in INPUT_SCALARS:
        if f == "horizon":
            continue
        v = data["in." + f]
        if f in ("dt", "min_rel_cost_change", "mu"):
            setattr(opt, f, float(v))
        elif f == "use_quadratic_terms":
            setattr(opt, f, bool(v))
        else:
            setattr(opt, f, int(v))
    for f in INPUT_ARRAYS:
        setattr(opt, f, data["in." + f])
    for key in data.files if hasattr(data, "files") else data.keys():
        if key.startswith("in.param."):
            v = data[key]
            setattr(opt.params, key[len("in.param."):], float(v) if v.ndim == 0 else v)


def run_sequence(opt, shift):
    """update; warm-start (shift + perturb x0); update; update.  Returns the snapshots."""
    snaps = []
    opt.update()
    snaps.append(snapshot(opt))
    if shift:
        opt.shift(shift)
    x0 = np.array(opt.x[0], dtype=np.float64, copy=True)
    x0[0] += 0.01
    opt.x[0] = x0
    opt.update()
    snaps.append(snapshot(opt))
    opt.update()
    snaps.append(snapshot(opt))
    return snaps


# ---------------------------------------------------------------------------
# comparison against a golden file
# ---------------------------------------------------------------------------
# The reference solver is built -O3 -ffast-math, so bitwise equality is not
# defined.  Tolerances are relative to the largest magnitude of the array
# ("scale"), which is how "trajectories within 1e-5 rel of reference" is read.
TRAJ_FIELDS = ["x", "u", "lagrange_multiplier", "next_x", "next_u", "int_step"]
DERIV_FIELDS = ["fx", "fu", "lx", "lu", "lxx", "luu", "lux", "K"]
DISCRETE_FIELDS = ["iterations", "lg_iterations", "mu_step", "termination_condition", "improved",
                   "trajectory_changed"]


def scaled_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
    if ref.size == 0:
        return 0.0
    scale = max(float(np.max(np.abs(ref))), 1e-12)
    return float(np.max(np.abs(got - ref))) / scale


def at_noise_floor(golden, i):
    """True if update i of the golden sequence ended on a cost plateau, where the
    accept/reject decision of the last line search is decided by rounding."""
    if i == 0:
        return False
    j0 = float(golden[f"out{i-1}.traj_costs"])
    j1 = float(golden[f"out{i}.traj_costs"])
    return abs(j1 - j0) <= 1e-12 * abs(j1)


def compare(snaps, golden, traj_tol=1e-6, cost_tol=1e-8, deriv_tol=1e-6, first_order=False):
    """Returns a list of human-readable mismatches (empty = parity)."""
    bad = []
    for i, s in enumerate(snaps):
        floor = at_noise_floor(golden, i)
        for f in TRAJ_FIELDS:
            if floor and f in ("next_x", "next_u"):
                continue
            e = scaled_err(s[f], golden[f"out{i}.{f}"])
            if e > traj_tol:
                bad.append(f"update {i}: {f} differs by {e:.2e} of scale (tol {traj_tol:g})")
        jr = float(golden[f"out{i}.traj_costs"])
        if abs(float(s["traj_costs"]) - jr) > cost_tol * abs(jr):
            bad.append(f"update {i}: traj_costs {float(s['traj_costs'])!r} vs {jr!r}")
        derivs = [f for f in DERIV_FIELDS if not (first_order and f in ("lxx", "luu", "lux", "K"))]
        # later updates start from an already converged plan: rounding noise of the
        # finite-differenced Hessians decides more of the result there
        dtol = deriv_tol if i == 0 else 10.0 * deriv_tol
        for f in derivs:
            e = scaled_err(s[f], golden[f"out{i}.{f}"])
            # the feedback gains divide by Quu, whose entries carry the rounding noise of the
            # finite-differenced Hessians (x 1e8): they get the looser bound from the start
            if e > (10.0 * deriv_tol if f == "K" else dtol):
                bad.append(f"update {i}: {f} differs by {e:.2e} of scale (tol {dtol:g})")
        if not floor:
            for f in DISCRETE_FIELDS:
                if float(s[f]) != float(golden[f"out{i}.{f}"]):
                    bad.append(f"update {i}: {f} = {float(s[f])} vs {float(golden[f'out{i}.{f}'])}")
            for f in ("alpha", "mu"):
                if abs(float(s[f]) - float(golden[f"out{i}.{f}"])) > 1e-9 * abs(float(golden[f"out{i}.{f}"])):
                    bad.append(f"update {i}: {f} = {float(s[f])} vs {float(golden[f'out{i}.{f}'])}")
        else:
            if int(s["iterations"]) != int(golden[f"out{i}.iterations"]):
                bad.append(f"update {i}: iterations = {int(s['iterations'])} vs {int(golden[f'out{i}.iterations'])}")
    return bad
