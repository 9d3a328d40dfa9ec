#!/usr/bin/env python3
"""Generates tests/golden/ilqr_<case>.npz by running the REFERENCE genopt solver
(imported from /root/reference, which only exists in the build container; the
reference itself never travels -- only these input/output vectors do).

    CFLAGS="-DPyArray_MoveInto=PyArray_CopyInto" python tests/golden/make_ilqr_goldens.py

(NumPy 2 removed PyArray_MoveInto, which the reference template calls at
optim.c:1343; the reference pins numpy 1.23.  The define leaves the reference
untouched; PyArray_CopyInto has the same effect for non-overlapping arrays.)

For every case of tests/ilqr_cases.py: inputs after set-up ("in.*"), outputs
after each of three consecutive update() calls ("out0.* .. out2.*"), and a few
dynamics()/ct_dynamics() values ("dyn.*").
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/library")
os.environ.setdefault("CFLAGS", "-DPyArray_MoveInto=PyArray_CopyInto")

import ilqr_cases as cases  # noqa: E402
from tpl.optim import genopt, optimizers as ref_opts  # noqa: E402


def reference_factory(model):
    if model == "simple_vehicle":
        import sympy as sp
        x, y, heading, v, steer = sp.symbols("x y heading v steer")
        x_trg, y_trg, w_e_pos, w_e_v, w_steer = sp.symbols("x_trg y_trg w_e_pos w_e_v w_steer")
        miss = w_e_pos * (x - x_trg)**2 + w_e_pos * (y - y_trg)**2
        cfg = genopt.Config([x, y, heading], [v, steer], [x_trg, y_trg, w_e_pos, w_e_v, w_steer],
                            sp.Matrix([v * sp.cos(heading), v * sp.sin(heading), v * steer]),
                            miss + w_e_v * v**2 + w_steer * steer**2, miss, use_cache=True)
    elif model == "builtin_probe":
        # same problem statement as tpl_amd.optim.optimizers.config_builtin_probe, written with the
        # reference's own symext classes (its generator keys on their types)
        import sympy as sp
        from tpl.optim import symext as rspx
        t, dt, x, y, u = sp.symbols("t dt x y u")
        w_f, w_u, x0g, y0g, dxg, dyg, loop_len, loop_dx = sp.symbols("w_f w_u x0g y0g dxg dyg loop_len loop_dx")
        field, loop_xs, loop_vals, steps, gains = (rspx.ArraySymbol(n) for n in "field loop_xs loop_vals steps gains".split())
        f_xy = rspx.blerp(x0g, y0g, dxg, dyg, x, y, field)
        drift = rspx.lerp_wrap(loop_len, loop_dx, x, loop_xs, loop_vals)
        gain = rspx.box_interp(dt, t * dt, steps) * rspx.get_array_value(gains, t)
        cfg = genopt.Config([x, y], [u],
                            [w_f, w_u, x0g, y0g, dxg, dyg, loop_len, loop_dx, field, loop_xs, loop_vals, steps, gains],
                            sp.Matrix([sp.cos(y) + drift, gain * u]), w_f * f_xy**2 + w_u * u**2,
                            end_costs=w_f * f_xy**2, constraints=[y - 2.0], use_cache=True)
    else:
        cfg = getattr(ref_opts, "config_" + model)()
    cfg.output_dir = "/tmp/genopt_oracle/"
    return genopt.build(cfg)


def main():
    factories = {}
    only = set(sys.argv[1:])
    for name, (model, setup, kw, shift) in cases.CASES.items():
        if only and name not in only:
            continue
        if model not in factories:
            factories[model] = reference_factory(model)
        opt = factories[model]()
        setup(opt, **kw)
        data = {}
        for k, v in cases.snapshot(opt, cases.INPUT_ARRAYS).items():
            data["in." + k] = v
        for k, v in cases.param_snapshot(opt).items():
            data["in." + k] = v
        data["in.shift"] = np.int64(shift)
        # dynamics spot values at the initial state (optim.c:1496-1652)
        x0 = np.array(opt.x[0], dtype=np.float64)
        u0 = np.array(opt.u[0], dtype=np.float64).reshape(-1)
        data["dyn.x"] = x0
        data["dyn.u"] = u0 + 0.1
        data["dyn.dynamics"] = opt.dynamics(x0, u0 + 0.1, 3, 0.07)
        data["dyn.ct_dynamics"] = opt.ct_dynamics(x0, u0 + 0.1, 3, 0.07)
        snaps = cases.run_sequence(opt, shift)
        for i, s in enumerate(snaps):
            for k, v in s.items():
                data[f"out{i}.{k}"] = v
        path = os.path.join(HERE, f"ilqr_{name}.npz")
        np.savez_compressed(path, **data)
        print(f"{name:28s} J={[float(s['traj_costs']) for s in snaps]} it={[int(s['iterations']) for s in snaps]} "
              f"term={[int(s['termination_condition']) for s in snaps]} -> {os.path.getsize(path)//1024} KiB")


if __name__ == "__main__":
    main()
