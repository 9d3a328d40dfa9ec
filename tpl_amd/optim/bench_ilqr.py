"""Batched iLQR measurement (BASELINE configs[2]): 1024 random initial states,
kinematic single-track tracking MPC (trajectory_tracking_mpc, X=7 U=2 C=4),
HEUN, N = 50, dt = 0.05, 20 iterations per update with min_rel_cost_change = 0
(every problem runs every iteration), device-resident.  One "batched iteration"
= one pass of the iLQR loop body (derivative refresh + backward pass + line
search) for every problem of the batch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _configure(opt, x0):
    """ModelPredictiveController's set-up (control/model_predictive_controller.py:69-76,
    134-155) with the cost weights of data/params/control/default/state.json."""
    opt.integrator_type = opt.HEUN
    opt.lg_mult_limit = 0.0
    opt.step = 0.05
    opt.max_iterations = 20
    opt.max_lg_iterations = 1
    opt.min_rel_cost_change = 0.0
    opt.barrier_weight[:] = 10000.0
    opt.lagrange_multiplier[:] = 0.0
    opt.u_min[..., 0] = -3.0
    opt.u_max[..., 0] = 1.5
    opt.u_min[..., 1] = -1.0
    opt.u_max[..., 1] = 1.0
    n = 100
    s = np.arange(n) * 0.5
    p = opt.params
    p.v_ch, p.max_delta, p.min_acc, p.max_acc, p.l, p.ref_step, p.a_offset = 32.0, 0.7, -3.0, 2.0, 3.165, 0.5, 0.0
    p.ref_x, p.ref_y, p.ref_phi, p.ref_k, p.ref_v = s, np.zeros(n), np.zeros(n), np.zeros(n), np.full(n, 8.0)
    for k, v in dict(pd=10.0, pv=5.0, pdelta=0.0, min_pdelta_dot=0.1, pdelta_dot=0.0, min_p_phi_dot=0.0,
                     p_phi_dot=0.05, p_phi=1000.0, p_phi_ref_dot_diff=0.01, pa=2.0, pj=0.5).items():
        setattr(p, k, v)
    opt.x[..., 0, :] = x0
    opt.u[:] = 0.0


def initial_states(batch, seed=0):
    rng = np.random.default_rng(seed)
    x0 = np.zeros((batch, 7))
    x0[:, 1] = rng.uniform(-1, 1, batch)
    x0[:, 2] = rng.uniform(-0.2, 0.2, batch)
    x0[:, 4] = rng.uniform(4, 10, batch)
    return x0


def run(device=0, batch=1024, horizon=50, updates=10, warmup=2, cpu_budget_s=10.0, seed=0, barrier=None,
        id_offset=0):
    """One rank's share of the batched-iLQR workload.  `seed` picks this rank's block of initial states,
    `barrier` (N > 1) brackets the timed region, `id_offset` is the global id of this rank's problem 0."""
    from . import genopt, optimizers as o
    make = genopt.build(o.config_trajectory_tracking_mpc(), "trajectory_tracking_mpc")
    opt = make(batch=batch, horizon=horizon, device=device)
    x0 = initial_states(batch, seed)
    _configure(opt, x0)
    opt.upload()
    for _ in range(warmup):
        opt.restore()
        opt.run_resident()
    opt.sync()
    opt.timing(reset=True)
    if barrier is not None:
        barrier()
    t0 = time.perf_counter()
    for _ in range(updates):
        opt.restore()
        opt.run_resident()
    opt.sync()
    if barrier is not None:
        barrier()
    el = time.perf_counter() - t0
    kernel_ms, launches = opt.timing(reset=True)
    opt.download()
    iters = int(opt.iterations[0])
    out = {
        "metric": "batched_ilqr_iters_per_s",
        "value": updates * iters / el,
        "unit": "batched iLQR iterations/s (1 iteration = all problems of the batch)",
        "problem_iterations_per_s": updates * iters * batch / el,
        "config": {"workload": f"trajectory_tracking_mpc X=7 U=2 C=4, HEUN, N={horizon}, dt=0.05, batch={batch}, "
                               f"{iters} iterations/update, min_rel_cost_change=0", "batch": batch, "N": horizon},
        "dtype": "f64",
        "ms_per_update": el / updates * 1e3,
        "kernel_ms_per_update": kernel_ms / max(1, launches),
        "all_finite": bool(np.all(np.isfinite(opt.traj_costs))),
        "elapsed_s": el, "updates": updates, "iterations_per_update": iters,
    }
    # this rank's best problem as the fixed-size record of SURVEY 8e: {constr, cost, id, x[T+1,X], u[T,U]}
    b = int(np.argmin(np.where(np.isfinite(opt.traj_costs), opt.traj_costs, np.inf)))
    out["best_record"] = np.concatenate([[0.0, float(opt.traj_costs[b]), float(id_offset + b)],
                                         np.asarray(opt.x[b], dtype=np.float64).ravel(),
                                         np.asarray(opt.u[b], dtype=np.float64).ravel()])
    if cpu_budget_s > 0:
        out["cpu_baseline"] = cpu_baseline(x0, horizon, cpu_budget_s)
    return out


def cpu_baseline(x0, horizon, budget_s):
    """The CPU oracle (port) on a bounded sample of the same problems, all host
    cores (OpenMP over problems); reset + solve are timed inside the C library."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_ilqr as OI
    lib = OI.load("trajectory_tracking_mpc")
    cores = int(lib.ilqo_max_threads())
    n = len(x0)
    opts = []
    for b in range(n):
        r = OI.OracleOptim("trajectory_tracking_mpc", t_cap=horizon)
        r.horizon = horizon
        _configure(r, x0[b])
        opts.append(r)
    hs = (C.c_void_p * n)(*[r._h for r in opts])
    x0c = np.ascontiguousarray(x0, dtype=np.float64)
    px = x0c.ctypes.data_as(C.POINTER(C.c_double))
    lib.ilqo_reset_and_update_many(hs, n, px, 1, 0)              # warm-up, also calibrates
    t0 = time.perf_counter()
    lib.ilqo_reset_and_update_many(hs, n, px, 1, 0)
    one = time.perf_counter() - t0
    reps = max(1, min(200, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    lib.ilqo_reset_and_update_many(hs, n, px, reps, 0)
    el = time.perf_counter() - t0
    iters = opts[0].iterations
    return {"value": reps * iters / el, "unit": "batched iLQR iterations/s (1024-problem batch)",
            "problem_iterations_per_s": reps * n * iters / el, "cores": cores, "kind": "port",
            "sample": f"{reps} x the same {n} problems x {iters} iterations "
                      f"(oracle/ilqr_oracle.cpp, OpenMP over problems)"}


if __name__ == "__main__":
    import json
    sys.path.insert(0, ROOT)
    res = run(cpu_budget_s=0.0 if "--no-cpu-baseline" in sys.argv else 10.0)
    res.pop("best_record", None)
    print(json.dumps(res))
