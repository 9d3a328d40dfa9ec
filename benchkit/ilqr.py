"""Batched iLQR measurement, GPU leg only (the CPU-oracle baseline lives in bench.py: nothing under tpl_amd/
touches oracle/ or tests/).

* ``mpc`` (BASELINE configs[2]): 1024 random initial states, kinematic single-track tracking MPC
  (trajectory_tracking_mpc, X=7 U=2 C=4), HEUN, N = 50, dt = 0.05, 20 iterations per update with
  min_rel_cost_change = 0 (every problem runs every iteration), device-resident.
* ``lateral`` (SURVEY 8d "spatial representative"): lateral_profile (X=2 U=1 C=2), EULER, T = 250,
  ds = 0.5, 5 iterations per update, as PathOptim runs it (planning/path_vel_decomp/path_optim.py:292-299).

One "batched iteration" = one pass of the iLQR loop body (derivative refresh + backward pass + line search)
for every problem of the batch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FP64_VECTOR_PEAK_TFLOPS = 78.6          # MI355X_MICROARCH.md / SURVEY 8d: fp64 vector = fp64 MFMA peak

# backward-pass operations per stage counted from optim.c:926-985
F_BW = {(7, 2): 2427, (6, 2): 1718, (2, 1): 118, (3, 1): 261, (4, 1): 486}


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


def _configure_lateral(opt, x0, horizon):
    """PathOptim's set-up of lateral_profile (path_optim.py:101-104, 125-133, 292-299): a corridor with an
    obstacle bump on the right, min_rel_cost_change = 0 so that every problem runs all 5 iterations."""
    opt.integrator_type = opt.EULER
    opt.lg_mult_limit = 0.0
    opt.barrier_weight[:] = 1000.0
    opt.step = 0.5
    opt.max_iterations = 5
    opt.max_lg_iterations = 1
    opt.min_rel_cost_change = 0.0
    opt.lagrange_multiplier[:] = 0.0
    opt.u_min[:] = -0.5
    opt.u_max[:] = 0.5
    s = np.arange(horizon) * 0.5
    p = opt.params
    p.ref_step, p.w_d, p.w_v_d, p.w_a_d, p.w_k = 0.5, 1.0, 10.0, 100.0, 50.0
    p.k_ref = 0.01 * np.sin(s / 20.0)
    p.d_offset = np.zeros(horizon)
    p.d_lower_constr = -2.0 + 2.6 * np.exp(-((s - 30.0) / 6.0) ** 2)
    p.d_upper_constr = np.full(horizon, 2.5)
    opt.x[..., 0, :] = x0
    opt.u[:] = 0.0


def initial_states_lateral(batch, seed=0):
    rng = np.random.default_rng(200 + seed)
    x0 = np.zeros((batch, 2))
    x0[:, 0] = rng.uniform(-0.5, 0.5, batch)
    return x0


WORKLOADS = {
    "mpc": dict(problem="trajectory_tracking_mpc", config="config_trajectory_tracking_mpc", horizon=50, dyn_evals=2,
                what="trajectory_tracking_mpc X=7 U=2 C=4, HEUN, N={T}, dt=0.05"),
    "lateral": dict(problem="lateral_profile", config="config_lateral_profile", horizon=250, dyn_evals=1,
                    what="lateral_profile X=2 U=1 C=2, EULER, T={T}, ds=0.5"),
}


def flops_per_stage(problem, dims, dyn_evals):
    """(F_bw, F_deriv, F_roll) per stage: backward-pass count of SURVEY 8a, and the generator's own operation
    counts of the routines one derivative refresh / one rollout stage evaluates (genopt.routine_flops)."""
    from tpl_amd.optim import genopt
    fl = genopt.routine_flops(problem)
    X, U = dims
    f = lambda name: fl.get(name, {}).get("flops", 0)
    f_deriv = (f("jacobian") + f("stateJacobian") + f("actionJacobian") + f("stateGradient") + f("actionGradient")
               + f("stateStateHessian") + f("actionActionHessian") + f("actionStateHessian"))
    # one rollout stage: the integrator's ctDynamics calls (+ ~4X flops of combination per call), the feedback law
    # u + alpha k + K (x' - x) with its clamp (2UX + 4U), the stage cost
    f_roll = dyn_evals * (f("ctDynamics") + 4 * X) + 2 * U * X + 4 * U + f("costs")
    return F_BW.get((X, U), 0), f_deriv, f_roll


def issue_block(roofline, problem, form, batch):
    """Share of the issued instructions that are fp64 arithmetic, from the committed PMC pass of this kernel
    (profiles/r0N_ilqr_issue.json, written by tools/profile_round.sh from rocprofv3 --pmc SQ_INSTS_VALU_*_F64 / SQ_INSTS_*;
    rocprofv3 cannot run inside the timed process): a constant of the repository, keyed by problem / launch form."""
    import json
    for f in ("r06_ilqr_issue.json", "r05_ilqr_issue.json", "r04_ilqr_issue.json"):
        try:
            with open(os.path.join(ROOT, "profiles", f)) as fd:
                tab = json.load(fd)
        except OSError:
            continue
        row = tab.get(f"{problem}/{form or 'single'}")
        if row:
            roofline["fp64_issue_share"] = row["fp64_issue_share"]
            roofline["fp64_issue_share_source"] = f"profiles/{f} ({row.get('what', 'PMC pass')}; kernel revision {row.get('profiled_kernel_rev', '?')}, batch {row.get('batch', '?')})"
            return
    roofline["fp64_issue_share"] = None


def run(device=0, batch=1024, horizon=None, updates=10, warmup=2, seed=0, barrier=None, id_offset=0, workload="mpc", lib=None):
    """One rank's share of a batched-iLQR workload.  `seed` picks this rank's block of initial states,
    `barrier` (N > 1) brackets the timed region, `id_offset` is the global id of this rank's problem 0."""
    from tpl_amd.optim import genopt, optimizers as o
    w = WORKLOADS[workload]
    horizon = horizon or w["horizon"]
    if lib:            # an experimental build of the same problem (tools/ab_*.py): never installed over the shipped library
        from tpl_amd.optim import solver
        make = solver.factory(lib, getattr(o, w["config"])().param_defaults)
    else:
        make = genopt.build(getattr(o, w["config"])(), w["problem"])
    opt = make(batch=batch, horizon=horizon, device=device)
    if workload == "mpc":
        x0 = initial_states(batch, seed)
        _configure(opt, x0)
    else:
        x0 = initial_states_lateral(batch, seed)
        opt.horizon = horizon
        _configure_lateral(opt, x0, horizon)
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
    iters = int(np.ravel(opt.iterations)[0])
    X, U = opt.x.shape[-1], (opt.u.shape[-1] if opt.u.ndim == 3 else 1)
    f_bw, f_deriv, f_roll = flops_per_stage(w["problem"], (X, U), w["dyn_evals"])
    # line-search tries of the reference's accept-first search on the LAST iteration: alpha = 10^-(tries-1)
    alpha = np.clip(np.ravel(np.asarray(opt.alpha, dtype=np.float64)), 1e-8, 1.0)
    n_ls = float(np.mean(np.rint(-np.log10(alpha)) + 1.0))
    per_iter = batch * horizon * (f_bw + f_deriv + n_ls * f_roll)
    per_iter_exec = batch * horizon * (f_bw + f_deriv + 8 * f_roll)          # the kernel rolls all 8 candidates out
    k_s = (kernel_ms / max(1, launches)) * 1e-3
    out = {
        "metric": "batched_ilqr_iters_per_s",
        "value": updates * iters / el,
        "unit": "batched iLQR iterations/s (1 iteration = all problems of the batch)",
        "problem_iterations_per_s": updates * iters * batch / el,
        "config": {"workload": (w["what"].format(T=horizon) + f", batch={batch}, {iters} iterations/update, "
                                "min_rel_cost_change=0"), "batch": batch, "N": horizon},
        "dtype": "f64",
        "ms_per_update": el / updates * 1e3,
        "kernel_ms_per_update": kernel_ms / max(1, launches),
        "all_finite": bool(np.all(np.isfinite(opt.traj_costs))),
        "form": opt.info().get("form"),
        "elapsed_s": el, "updates": updates, "iterations_per_update": iters,
        # algorithmic flops per batched iteration = batch x T x (F_bw + F_deriv + n_ls F_roll) over the
        # kernel time of one iteration, against the fp64 vector peak.  The path is latency-bound by construction
        # (50-250 dependent stages on one wavefront per problem); MFMA is not used.
        "roofline": {"bound": "valu_fp64", "unit": "TFLOP/s", "peak": FP64_VECTOR_PEAK_TFLOPS,
                     "achieved": per_iter * iters / k_s / 1e12, "frac": per_iter * iters / k_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                     "traffic": None, "kernel": {"split": "k_split_chain + k_split_refresh", "fused": "k_ilqr_fused"}.get(opt.info().get("form"), "k_ilqr_update"),
                     "flops_per_stage": {"F_bw": f_bw, "F_deriv": f_deriv, "F_roll": f_roll, "n_ls": n_ls},
                     "algorithmic_flops_per_batched_iteration": per_iter,
                     "executed_flops_per_batched_iteration": per_iter_exec,
                     "executed_tflops": per_iter_exec * iters / k_s / 1e12,
                     "mfma_busy": 0.0},
    }
    # What a caller of the drop-in boundary waits for: Optim.update() = upload of the host mirrors the caller wrote (x0, u,
    # bounds, multipliers, array parameters), the same launch, eager fetch of x / u / costs into the numpy views
    # (ilqr_model.hip: tplhip_ilqr_update) -- next to the device-resident figure above, same problems, cold start each time.
    call = []
    for _ in range(3):
        if workload == "mpc":
            _configure(opt, x0)
        else:
            _configure_lateral(opt, x0, horizon)
        t0 = time.perf_counter()
        opt.update()
        call.append((time.perf_counter() - t0) * 1e3)
    out["update_call_ms"] = float(np.median(call))
    out["update_call_batched_iters_per_s"] = iters / (out["update_call_ms"] * 1e-3)
    issue_block(out["roofline"], w["problem"], out["form"], batch)
    # this rank's best problem as the fixed-size record of SURVEY 8e: {constr, cost, id, x[T+1,X], u[T,U]}
    b = int(np.argmin(np.where(np.isfinite(opt.traj_costs), opt.traj_costs, np.inf)))
    out["best_record"] = np.concatenate([[0.0, float(opt.traj_costs[b]), float(id_offset + b)],
                                         np.asarray(opt.x[b], dtype=np.float64).ravel(),
                                         np.asarray(opt.u[b], dtype=np.float64).ravel()])
    out["x0"] = x0
    return out


def batch_sweep(device=0, batches=(256, 512, 1024, 2048, 4096, 8192), updates=3, warmup=2, workload="mpc"):
    """Problem-iterations/s against the batch size: 1024 problems are one wavefront per SIMD of the chip; beyond that
    update() takes the fused form (the phases as calls inside one launch, two wavefronts per SIMD resident; `form`)."""
    out = []
    for b in batches:
        r = run(device=device, batch=b, updates=updates, warmup=warmup, workload=workload)
        out.append({"batch": b, "form": r["form"], "batched_iters_per_s": r["value"],
                    "problem_iterations_per_s": r["problem_iterations_per_s"], "kernel_ms_per_update": r["kernel_ms_per_update"],
                    "fp64_frac": r["roofline"]["frac"]})
    return out


CROSSOVER = {"mpc": dict(horizon=20, iterations=5), "lateral": dict(horizon=250, iterations=5)}


def configure_crossover(opt, workload, x0):
    """The problem of the crossover table: the caller's shape (ModelPredictiveController plans N = 20 stages,
    PathOptim T = 250 stations), 5 iterations per update() -- the template's default, min_rel_cost_change = 0 so that
    the CPU and the GPU leg do exactly the same work -- from a cold start."""
    c = CROSSOVER[workload]
    if workload == "mpc":
        opt.horizon = c["horizon"]
        _configure(opt, x0)
    else:
        opt.horizon = c["horizon"]
        _configure_lateral(opt, x0, c["horizon"])
    opt.max_iterations = c["iterations"]
    opt.min_rel_cost_change = 0.0


def crossover_gpu(device=0, workload="mpc", batches=(1, 2, 4, 8, 16, 32, 64), reps=30):
    """Wall time of ONE synchronous update() -- mirrors up, kernel, mirrors down: what a caller of the plugin surface
    waits for -- against the batch size.  With bench.py's one-core CPU figure for the same problem this gives the batch
    from which the HIP solver is the faster way to solve a set of problems."""
    from tpl_amd.optim import genopt, optimizers as o
    w = WORKLOADS[workload]
    make = genopt.build(getattr(o, w["config"])(), w["problem"])
    rows = []
    for b in batches:
        opt = make(batch=b, horizon=CROSSOVER[workload]["horizon"], device=device) if b > 1 else make(device=device)
        x0 = initial_states(b) if workload == "mpc" else initial_states_lateral(b)
        lat = []
        for r in range(reps + 3):
            configure_crossover(opt, workload, x0 if b > 1 else x0[0])
            t0 = time.perf_counter()
            opt.update()
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.array(lat[3:])
        rows.append({"batch": b, "update_ms_p50": float(np.percentile(lat, 50)), "ms_per_problem": float(np.percentile(lat, 50)) / b,
                     "iterations": int(np.ravel(opt.iterations)[0])})
        del opt
    return rows


if __name__ == "__main__":
    import json
    sys.path.insert(0, ROOT)
    if "--batch-sweep" in sys.argv:
        print(json.dumps(batch_sweep()))
        sys.exit(0)
    res = run(workload=sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "mpc")
    res.pop("best_record", None)
    res.pop("x0", None)
    print(json.dumps(res))
