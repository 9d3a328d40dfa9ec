"""The optimal-control problems tpl ships (mirror of the reference's
``tpl.optim.optimizers``, library/tpl/optim/optimizers.py).

Each ``config_*`` function returns a :class:`genopt.Config`; the problem
statements (states, actions, parameter order, dynamics, costs, constraints) are
those of the reference, cited per function, because they define the numbers the
solvers must reproduce.  ``build_optimizers`` generates and compiles all of
them and publishes the factories as module attributes like the reference does
(optimizers.py:560-582).
"""

import sympy as sp

from . import genopt
from . import symext as spx

S = sp.Symbol
A = spx.ArraySymbol


def _symbols(names):
    return [S(n) for n in names.split()]


def _single_track_yaw_rate(v, delta, l, v_ch, extra=1):
    """Yaw rate of the kinematic single-track model with the characteristic
    velocity v_ch (understeer gradient)."""
    return v / (l * (1 + (v / v_ch)**2)) * extra * sp.tan(delta)


def _steer_acc_limits(delta, a, max_delta, max_acc, min_acc):
    return [delta - max_delta, -max_delta - delta, a - max_acc, min_acc - a]


def config_trajectory_tracking_mpc():
    """Path-tracking MPC, X=7 [x,y,phi,delta,v,s_r,a], U=2 [j,delta_dot], C=4
    (optimizers.py:12-126)."""
    x, y, phi, delta, v, s_r, a = _symbols("x y phi delta v s_r a")
    j, delta_dot = _symbols("j delta_dot")
    ref_x, ref_y, ref_phi, ref_k, ref_v = (A(n) for n in "ref_x ref_y ref_phi ref_k ref_v".split())
    ref_step, l, v_ch, max_delta, max_acc, min_acc, a_offset = _symbols(
        "ref_step l v_ch max_delta max_acc min_acc a_offset")
    (pd, pv, pdelta, pdelta_dot, p_phi_dot, min_p_phi_dot, min_pdelta_dot,
     p_phi, p_phi_ref_dot_diff, pa, pj) = _symbols(
        "pd pv pdelta pdelta_dot p_phi_dot min_p_phi_dot min_pdelta_dot p_phi p_phi_ref_dot_diff pa pj")

    # reference course at the tracked arc length s_r
    r_x = spx.lerp(0.0, ref_step, s_r, ref_x)
    r_y = spx.lerp(0.0, ref_step, s_r, ref_y)
    r_phi = spx.lerp_angle(0.0, ref_step, s_r, ref_phi)
    r_k = spx.lerp(0.0, ref_step, s_r, ref_k)
    v_trg = spx.lerp(0.0, ref_step, spx.fixed(s_r), ref_v)      # not differentiated through

    phi_dot = _single_track_yaw_rate(v, delta, l, v_ch)
    d_r = sp.cos(r_phi) * (y - r_y) - sp.sin(r_phi) * (x - r_x)  # lateral offset to the course
    s_dot = v * sp.cos(phi - r_phi) / (1 - d_r * r_k)            # progress along the course

    dynamics = sp.Matrix([v * sp.cos(phi), v * sp.sin(phi), phi_dot, delta_dot, a + a_offset, s_dot, j])

    costs = 0.0
    costs += (min_pdelta_dot + pdelta_dot * v**2) * delta_dot**2
    costs += (min_p_phi_dot + p_phi_dot * v**2) * phi_dot**2
    costs += pa * a**2
    costs += pj * j**2
    costs += pv * (v - v_trg)**2
    costs += pd * d_r**2
    costs += p_phi * (1.0 - sp.cos(phi - r_phi))
    costs += p_phi_ref_dot_diff * (phi_dot - s_dot * r_k)**2 * v**2

    return genopt.Config(
        [x, y, phi, delta, v, s_r, a],
        [j, delta_dot],
        [pd, pv, pdelta, min_pdelta_dot, pdelta_dot, min_p_phi_dot, p_phi_dot, p_phi,
         p_phi_ref_dot_diff, pa, pj, l, v_ch, ref_x, ref_y, ref_phi, ref_k, ref_v, ref_step,
         max_delta, max_acc, min_acc, a_offset],
        dynamics, costs, end_costs=0.0,
        constraints=_steer_acc_limits(delta, a, max_delta, max_acc, min_acc),
        name="trajectory_tracking_mpc")


def config_trajectory_tracking_mpc_time():
    """Time-indexed tracking MPC, X=6 [x,y,phi,delta,v,a], U=2, C=4
    (optimizers.py:129-240)."""
    t, dt = _symbols("t dt")
    x, y, phi, delta, v, a = _symbols("x y phi delta v a")
    j, delta_dot = _symbols("j delta_dot")
    ref_x, ref_y, ref_phi, ref_v = (A(n) for n in "ref_x ref_y ref_phi ref_v".split())
    ref_dt, ref_t_offset, l, v_ch, max_delta, max_acc, min_acc, a_offset, cog_pos = _symbols(
        "ref_dt ref_t_offset l v_ch max_delta max_acc min_acc a_offset cog_pos")
    (pd, pv, pdelta, pdelta_dot, p_phi_dot, min_p_phi_dot, min_pdelta_dot, p_phi, pa, pj) = _symbols(
        "pd pv pdelta pdelta_dot p_phi_dot min_p_phi_dot min_pdelta_dot p_phi pa pj")

    rt = ref_t_offset + dt * t                                    # wall time of stage t
    r_x = spx.lerp(0.0, ref_dt, rt, ref_x)
    r_y = spx.lerp(0.0, ref_dt, rt, ref_y)
    r_phi = spx.lerp_angle(0.0, ref_dt, rt, ref_phi)
    v_trg = spx.lerp(0.0, ref_dt, rt, ref_v)

    beta = sp.atan(sp.tan(delta) * cog_pos)                       # side slip at the centre of gravity
    phi_dot = v * sp.tan(delta) * sp.cos(beta) / (l * (1 + (v / v_ch)**2))

    dynamics = sp.Matrix([v * sp.cos(phi + beta), v * sp.sin(phi + beta), phi_dot, delta_dot, a + a_offset, j])

    costs = 0.0
    costs += (min_pdelta_dot + pdelta_dot * v**2) * delta_dot**2
    costs += (min_p_phi_dot + p_phi_dot * v**2) * phi_dot**2
    costs += pa * a**2
    costs += pj * j**2
    costs += pv * (v - v_trg)**2
    costs += pd * (x - r_x)**2 + pd * (y - r_y)**2
    costs += p_phi * (1.0 - sp.cos(phi - r_phi))

    return genopt.Config(
        [x, y, phi, delta, v, a],
        [j, delta_dot],
        [pd, pv, pdelta, min_pdelta_dot, pdelta_dot, min_p_phi_dot, p_phi_dot, p_phi, pa, pj,
         l, v_ch, cog_pos, ref_x, ref_y, ref_phi, ref_v, ref_dt, ref_t_offset,
         max_delta, max_acc, min_acc, a_offset],
        dynamics, costs, end_costs=0.0,
        constraints=_steer_acc_limits(delta, a, max_delta, max_acc, min_acc),
        name="trajectory_tracking_mpc_time")


def config_lateral_profile():
    """Lateral path planning over arc length, X=2 [d, d'], U=1 [d''], C=2
    (optimizers.py:243-294)."""
    idx, ds = S("t"), S("dt")
    d, v_d, a_d = _symbols("d v_d a_d")
    k_ref, d_offset, d_lower_constr, d_upper_constr = (
        A(n) for n in "k_ref d_offset d_lower_constr d_upper_constr".split())
    ref_step, w_d, w_v_d, w_a_d, w_k = _symbols("ref_step w_d w_v_d w_a_d w_k")

    s = idx * ds
    k_r = spx.lerp(0.0, ref_step, s, k_ref)
    d_o = spx.lerp(0.0, ref_step, s, d_offset)
    d_lower = spx.lerp(0.0, ref_step, s, d_lower_constr)
    d_upper = spx.lerp(0.0, ref_step, s, d_upper_constr)

    # Cartesian curvature of the offset curve from its Frenet derivatives
    k = (a_d / (v_d**2 + 1) + k_r) * sp.cos(sp.atan(v_d)) / (1 - d * k_r)

    costs = w_d * (d - d_o)**2 + w_v_d * v_d**2 + w_a_d * a_d**2 + w_k * k**2
    end_costs = w_d * (d - d_o)**2 + w_v_d * v_d**2

    return genopt.Config(
        [d, v_d], [a_d],
        [k_ref, d_offset, d_lower_constr, d_upper_constr, ref_step, w_d, w_v_d, w_a_d, w_k],
        sp.Matrix([v_d, a_d]), costs, end_costs=end_costs,
        constraints=[d_lower - d, d - d_upper],
        name="lateral_profile")


def config_velocity_profile_time():
    """Velocity profile over time, X=2 [s,v], U=1 [a], C=4 (optimizers.py:297-349;
    defined by the reference but not in its build list)."""
    t = S("t")
    s, v, a = _symbols("s v a")
    w_v, w_a, ref_step = _symbols("w_v w_a ref_step")
    ref_v, ref_s_max, ref_s_min = A("ref_v"), A("ref_s_max"), A("ref_s_min")

    v_max = spx.lerp(0.0, ref_step, s, ref_v)
    s_max = spx.get_array_value(ref_s_max, t)
    s_min = spx.get_array_value(ref_s_min, t)

    constraints = [
        0.0 - v,
        v - v_max,
        sp.Piecewise((s - s_max, s_max > 0), (0.0, True)),
        sp.Piecewise((s_min - s, s_min > 0), (0.0, True)),
    ]
    costs = sp.Matrix([w_v * (1000 - v) + w_a * a**2])

    return genopt.Config(
        [s, v], [a], [w_v, w_a, ref_v, ref_step, ref_s_max, ref_s_min],
        sp.Matrix([v, a]), costs, end_costs=0.0, constraints=constraints,
        name="velocity_profile_time")


def config_velocity_profile_space():
    """Velocity profile over arc length, X=2 [v, st], U=1 [a], C=5, piecewise
    dynamics below 1 m/s (optimizers.py:352-428)."""
    t, dt = _symbols("t dt")
    st, v, a = _symbols("st v a")
    ref_step, p_v, p_a, max_a_total = _symbols("ref_step p_v p_a max_a_total")
    ref_t_offset, ref_v, ref_k, ref_t_max, ref_t_min, ref_v_weight = (
        A(n) for n in "ref_t_offset ref_v ref_k ref_t_max ref_t_min ref_v_weight".split())

    pos = t * dt
    t_offset = spx.box_interp(ref_step, pos, ref_t_offset)
    moving = v > 1.0 + 1e-3
    dynamics = sp.Matrix([
        sp.Piecewise((a / v, moving), (a, True)),
        sp.Piecewise((1.0 / v, moving), (t_offset, True)),
    ])

    v_trg = spx.lerp(0.0, ref_step, pos, ref_v)
    kk = spx.box_interp(ref_step, pos, ref_k)
    ost = st + t_offset
    t_min = spx.lerp(0.0, ref_step, pos, ref_t_min)
    t_max = spx.lerp(0.0, ref_step, pos, ref_t_max)
    a_lat = v**2 * kk
    v_weight = spx.lerp(0.0, ref_step, pos, ref_v_weight)

    constr_a_max = (a**2 + a_lat**2) - max_a_total**2
    constr_v_min = 1.0 - v
    constr_v_max = v - v_trg
    constr_t_max = ost - t_max
    constr_t_min = (t_min - st) * sp.Piecewise((v - 1.0, t_min > 0.0), (1.0, True))

    costs = sp.Matrix([p_v * (v_trg - v)**2 * v_weight + p_a * a**2])

    return genopt.Config(
        [v, st], [a],
        [p_v, p_a, max_a_total, ref_v, ref_k, ref_step, ref_t_max, ref_t_min, ref_t_offset, ref_v_weight],
        dynamics, costs, end_costs=0.0,
        constraints=[constr_a_max, constr_v_min, constr_v_max, constr_t_max, constr_t_min],
        name="velocity_profile_space")


def _ref_line_tracking_cost(x, y, s, w_pos, ref_step, ref_x, ref_y):
    return (w_pos * (x - spx.lerp(0.0, ref_step, s, ref_x))**2
            + w_pos * (y - spx.lerp(0.0, ref_step, s, ref_y))**2)


def config_ref_line_smoother_k():
    """Reference-line smoothing with curvature as action, X=3 [x,y,phi], U=1 [k]
    (optimizers.py:431-490)."""
    t, dt = _symbols("t dt")
    x, y, phi, k = _symbols("x y phi k")
    w_pos, w_k, ref_step = _symbols("w_pos w_k ref_step")
    ref_x, ref_y = A("ref_x"), A("ref_y")
    costs = _ref_line_tracking_cost(x, y, t * dt, w_pos, ref_step, ref_x, ref_y) + w_k * k**2
    return genopt.Config(
        [x, y, phi], [k], [w_pos, w_k, ref_x, ref_y, ref_step],
        sp.Matrix([sp.cos(phi), sp.sin(phi), k]), costs, end_costs=0.0,
        name="ref_line_smoother_k")


def config_ref_line_smoother_dk():
    """Reference-line smoothing with curvature rate as action, X=4 [x,y,phi,k],
    U=1 [dk] (optimizers.py:493-557)."""
    t, dt = _symbols("t dt")
    x, y, phi, k, dk = _symbols("x y phi k dk")
    w_pos, w_k, w_dk, s_start, ref_step = _symbols("w_pos w_k w_dk s_start ref_step")
    ref_x, ref_y = A("ref_x"), A("ref_y")
    costs = (_ref_line_tracking_cost(x, y, s_start + t * dt, w_pos, ref_step, ref_x, ref_y)
             + w_k * k**2 + w_dk * dk**2)
    return genopt.Config(
        [x, y, phi, k], [dk], [w_pos, w_k, w_dk, s_start, ref_x, ref_y, ref_step],
        sp.Matrix([sp.cos(phi), sp.sin(phi), k, dk]), costs, end_costs=0.0,
        name="ref_line_smoother_dk")


def config_simple_vehicle():
    """Unicycle-like demo problem, X=3 [x,y,heading], U=2 [v, steer], C=0
    (library/tpl/optim/examples/simple_vehicle/main.py:10-58); the small smoke
    model."""
    x, y, heading, v, steer = _symbols("x y heading v steer")
    x_trg, y_trg, w_e_pos, w_e_v, w_steer = _symbols("x_trg y_trg w_e_pos w_e_v w_steer")
    miss = w_e_pos * (x - x_trg)**2 + w_e_pos * (y - y_trg)**2
    return genopt.Config(
        [x, y, heading], [v, steer], [x_trg, y_trg, w_e_pos, w_e_v, w_steer],
        sp.Matrix([v * sp.cos(heading), v * sp.sin(heading), v * steer]),
        miss + w_e_v * v**2 + w_steer * steer**2, end_costs=miss,
        name="simple_vehicle")


def config_builtin_probe():
    """Not a tpl problem: a small OCP that exercises the interpolation builtins no shipped
    problem uses (``blerp``, ``lerp_wrap``) next to ``box_interp`` / ``get_array_value``, so
    their semantics (optim.c:357-373, 413-486) are pinned by a golden vector as well."""
    t, dt = _symbols("t dt")
    x, y, u = _symbols("x y u")
    w_f, w_u, x0g, y0g, dxg, dyg, loop_len, loop_dx = _symbols("w_f w_u x0g y0g dxg dyg loop_len loop_dx")
    field, loop_xs, loop_vals, steps, gains = A("field"), A("loop_xs"), A("loop_vals"), A("steps"), A("gains")
    f_xy = spx.blerp(x0g, y0g, dxg, dyg, x, y, field)                  # potential field over (x, y)
    drift = spx.lerp_wrap(loop_len, loop_dx, x, loop_xs, loop_vals)    # periodic drift along x
    gain = spx.box_interp(dt, t * dt, steps) * spx.get_array_value(gains, t)
    dynamics = sp.Matrix([sp.cos(y) + drift, gain * u])
    costs = w_f * f_xy**2 + w_u * u**2
    end_costs = w_f * f_xy**2
    return genopt.Config(
        [x, y], [u],
        [w_f, w_u, x0g, y0g, dxg, dyg, loop_len, loop_dx, field, loop_xs, loop_vals, steps, gains],
        dynamics, costs, end_costs=end_costs, constraints=[y - 2.0],
        name="builtin_probe")


# the six problems the reference builds at application start (optimizers.py:562-569)
CONFIG_FUNCTIONS = [
    config_trajectory_tracking_mpc,
    config_trajectory_tracking_mpc_time,
    config_lateral_profile,
    config_velocity_profile_space,
    config_ref_line_smoother_k,
    config_ref_line_smoother_dk,
]
# additionally pre-built here: the seventh definition and the smoke model
EXTRA_CONFIG_FUNCTIONS = [config_velocity_profile_time, config_simple_vehicle, config_builtin_probe]


def _name_of(fn):
    return fn.__name__[len("config_"):]


def build_optimizers(force_rebuild=False, extras=False):
    fns = CONFIG_FUNCTIONS + (EXTRA_CONFIG_FUNCTIONS if extras else [])
    if not force_rebuild:
        fns = [f for f in fns if _name_of(f) not in globals()]
    if not fns:
        return
    cfgs = [f() for f in fns]
    names = [_name_of(f) for f in fns]
    factories = genopt.build_parallel(cfgs, names)
    globals().update(dict(zip(names, factories)))
