"""Same-box A/B of the three update() forms (single launch / split sequence / fused) on tpl's time-based tracking MPC (trajectory_tracking_mpc_time, X=6 U=2 C=4,
HEUN, N = 40, 20 iterations per update, min_rel_cost_change 0): python tools/ab_ilqr_mpc_time.py [batches...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilqr_cases as cases  # noqa: E402  (the controller's set-up of this problem)
from tpl_amd.optim import genopt, optimizers as o  # noqa: E402


def run(batch, form):
    os.environ["TPLHIP_DEBUG"] = f"ilqr_split={form}"
    make = genopt.build(o.config_trajectory_tracking_mpc_time(), "trajectory_tracking_mpc_time")
    one = make()
    cases.mpc_time(one, seed=0)
    opt = make(batch=batch, horizon=40)
    for f in ("integrator_type", "lg_mult_limit", "horizon", "step", "max_iterations"):
        setattr(opt, f, getattr(one, f))
    opt.min_rel_cost_change = 0.0
    opt.max_iterations = 20
    opt.barrier_weight[:] = one.barrier_weight
    opt.lagrange_multiplier[:] = 0.0
    opt.u_min[:] = one.u_min
    opt.u_max[:] = one.u_max
    opt.params = one.params
    rng = np.random.default_rng(3)
    x0 = np.tile(np.array(one.x[0]), (batch, 1))
    x0[:, 1] += rng.uniform(-0.5, 0.5, batch)
    x0[:, 4] += rng.uniform(-2, 2, batch)
    opt.x[:, 0] = x0
    opt.u[:] = 0.0
    opt.upload()
    opt.restore(); opt.run_resident(); opt.sync()
    opt.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(3):
        opt.restore()
        opt.run_resident()
    opt.sync()
    el = time.perf_counter() - t0
    ms, n = opt.timing(reset=True)
    opt.download()
    assert np.all(np.isfinite(opt.traj_costs))
    return {"batch": batch, "form": opt.info()["form"], "problem_iterations_per_s": round(3 * 20 * batch / el), "kernel_ms_per_update": round(ms / max(1, n), 3)}


if __name__ == "__main__":
    for b in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
        for form in (0, 1, 2):
            print(json.dumps(run(b, form)), flush=True)
