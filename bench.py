#!/usr/bin/env python3
"""bench.py -- measures BASELINE.json's metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one complete DP plan on device-resident inputs: distance map,
t_steps-1 backward stage sweeps over the (s,v,a) grid, one forward rollout.
value = stage sweeps per second over the whole job (all ranks), timed over
exactly K steps between barrier + device synchronize on both sides, max over
ranks.  For N > 1 the plans (independent scenarios) are sharded across ranks,
no data-path collective: "scaling": "weak".

Also on the line: "roofline" (HBM, algorithmic 32 B/cell/sweep over the sweep
kernel's HIP-event time), "cpu_baseline" (the CPU oracle timed on a bounded
sample on rank 0), and "secondary" (batched iLQR iterations/s etc.).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL = 32            # read next float4 + write this float4


def env_cpu_baseline(reps=20):
    """The occupancy-pipeline oracle (oracle/dp_env_oracle.c, one thread) on the scene of
    latency.env_update -- the checker timed as a baseline, never the product."""
    import numpy as np
    import oracle_env as oe
    from tpl_amd import scenarios
    ep = scenarios.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = scenarios.env_scene("arc", seed=1, n_objects=10)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        o = oe.OracleEnv(ep)
        o.set_ref_line(rl, step)
        for g, st in objs:
            o.insert_geometry(g, st)
        o.update()
        t.append((time.perf_counter() - t0) * 1e3)
    return {"value": float(np.median(t)), "unit": "ms per update", "cores": 1, "kind": "port",
            "sample": f"{reps} updates of the same scene"}


def poly_lat_cpu_baseline(reps=10):
    """oracle/poly_lat_oracle.c (one thread) on the scene of latency.poly_lat_update."""
    import numpy as np
    import oracle_env as oe
    import oracle_poly_lat as opl
    import tpl_amd
    from tpl_amd import scenarios
    ep = scenarios.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = scenarios.env_scene("arc", seed=1, n_objects=10)
    o = oe.OracleEnv(ep)
    o.set_ref_line(rl, step)
    for g, st in objs:
        o.insert_geometry(g, st)
    _, occ, _ = o.update()
    start = tpl_amd.PolyLatTrajPoint()
    start.s, start.l, start.dl, start.v = 5.0, 0.4, 0.05, 8.0
    p = tpl_amd.PolyLatPlannerParams()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        opl.plan(p, o.c, start, o.ref, o.step, o.offsets, occ)
        t.append((time.perf_counter() - t0) * 1e3)
    return {"value": float(np.median(t)), "unit": "ms per update", "cores": 1, "kind": "port",
            "sample": f"{reps} updates of the same scene"}


def lat_lon_cpu_baseline(reps=3):
    """oracle/lat_lon_oracle.c (OpenMP over cells) on the scene and grid of latency.lat_lon_update."""
    import numpy as np
    import oracle_env as oe
    import oracle_lat_lon as oll
    import tpl_amd
    from tpl_amd import scenarios
    ep = scenarios.env_params(t_steps=10, s_steps=201, l_steps=31)
    rl, step, objs = scenarios.env_scene("arc", seed=1, n_objects=10)
    o = oe.OracleEnv(ep)
    o.set_ref_line(rl, step)
    for g, st in objs:
        o.insert_geometry(g, st)
    _, _, dist_lon = o.update()
    p = tpl_amd.DynProgLatLonPlannerParams()
    p.length_veh, p.width_veh = 4.9, 1.9
    start = tpl_amd.LatLonState()
    start.s, start.ds, start.l = 2.0, 8.0, 0.3
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        oll.update_traj_dp(p, o.c, dist_lon, o.ref, o.step, start)
        t.append((time.perf_counter() - t0) * 1e3)
    return {"value": float(np.median(t)), "unit": "ms per update_traj_dp", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{reps} plans of the same scene (OpenMP over cells)"}


def mpc_cpu_baseline(ticks=300, horizon=20):
    """oracle/ilqr_oracle.cpp (one thread, one problem) in the closed loop of latency.mpc_closed_loop:
    what the reference's CPU genopt solver costs per controller tick on this host."""
    import numpy as np
    import oracle_ilqr as OI
    from tpl_amd.optim import bench_ilqr
    opt = OI.OracleOptim("trajectory_tracking_mpc")
    opt.horizon = horizon
    x = np.array([0.0, 0.4, 0.05, 0.0, 6.0, 0.0, 0.0])
    bench_ilqr._configure(opt, x)
    opt.min_rel_cost_change = 1e-6
    opt.horizon = horizon
    lat = []
    for _ in range(ticks):
        opt.x[0] = x
        t0 = time.perf_counter()
        opt.update()
        lat.append((time.perf_counter() - t0) * 1e3)
        x = opt.dynamics(x, opt.u[0], 0, 0.01)
        x[5] = max(0.0, x[0])
        opt.shift(0)
    lat = np.array(lat[10:])
    return {"value": float(np.percentile(lat, 50)), "unit": "ms per update (p50)", "cores": 1, "kind": "port",
            "p99_ms": float(np.percentile(lat, 99)), "sample": f"{len(lat)} ticks of the same closed loop"}


def dp_setup(grid, device):
    import tpl_amd
    from tpl_amd import scenarios as sc
    S, V, A, T = sc.GRIDS[grid]
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
    env = tpl_amd.DynProgEnvironment(device)
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner(device)
    dp.reinit_buffers(p)
    dp.upload([sc.initial_state(0.0, 8.0, 0.0)], path)
    return dp, env, p, ep, path, occ


def dp_cpu_baseline(p, ep, path, occ, budget_s=12.0):
    """CPU oracle (port) on a bounded sample: whole stage sweeps of the same
    grid, all host cores, until ~budget_s of wall time is spent."""
    import ctypes as C
    import numpy as np
    import oracle_dp as O
    cp, cep = O.cparams(p), O.cparams(ep)
    dist = O.dist_map_path(cep, occ, path, cp.path_steps, cp.path_step_size)
    tf = int(O.lib().dpo_table_floats(C.byref(cp)))
    a = np.zeros(tf, dtype=np.float32)
    b = np.zeros(tf, dtype=np.float32)
    T = cp.t_steps
    O.lib().dpo_backward_stage(C.byref(cp), C.byref(cep), T - 1, O.fptr(a), O.fptr(a), O.fptr(path), O.fptr(dist), 0)
    n = 0
    t0 = time.perf_counter()
    while True:
        O.lib().dpo_backward_stage(C.byref(cp), C.byref(cep), T - 2, O.fptr(a), O.fptr(b), O.fptr(path), O.fptr(dist), 0)
        a, b = b, a
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    cores = int(O.lib().dpo_max_threads())
    return {"value": n / el, "unit": "stage-sweeps/s", "cores": cores, "kind": "port",
            "sample": f"{n} full stage sweeps of the same {cp.s_steps}x{cp.v_steps}x{cp.a_steps} grid "
                      f"(oracle/dp_lon_oracle.c, OpenMP over rows)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", default="target", help="target (256x128x64, default) | config2 | config4 | default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal hook for a one-GPU box: TPL_BENCH_REHEARSAL=1 runs the N > 1 code path with every rank on
    # device 0 and the all-gather over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = world > 1 and os.environ.get("TPL_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)

    dp, env, p, ep, path, occ = dp_setup(args.grid, local_rank)
    cells = p.s_steps * p.v_steps * p.a_steps
    sweeps_per_step = p.t_steps - 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from tpl_amd import sharding
    import numpy as np

    # N > 1: the path's only exchange is the best-plan record of every rank -> all ranks (RCCL
    # all-gather of 1.4 KB).  It is pipelined one step deep: the plan of step k is copied
    # out behind its kernels, and gathered while the GPU already runs step k+1.
    def exchange(slot):
        tr = np.array([[s.t, s.s, s.v, s.a, s.j, s.cost, s.constr] for s in dp.collect_download(slot)[0].states],
                      dtype=np.float32)
        sharding.allgather_best(sharding.plan_record(tr, rank), dist,
                                torch.device("cpu") if rehearsal else torch.device("cuda", local_rank))

    def run(n):
        for k in range(n):
            dp.plan_resident(env)
            if dist is not None:
                dp.stage_download(k & 1)
                if k > 0:
                    exchange((k - 1) & 1)
        if dist is not None and n > 0:
            exchange((n - 1) & 1)

    run(args.warmup)
    dp.sync()
    dp.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    dp.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    sweep_ms, sweep_launches = dp.timing_get()
    dp.timing_enable(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_sweeps = sweeps_per_step * args.steps * world
    value = total_sweeps / elapsed

    out = None
    if rank == 0:
        per_launch_s = (sweep_ms / 1e3) / max(1, sweep_launches)
        achieved = cells * BYTES_PER_CELL / per_launch_s / 1e9
        out = {
            "metric": "dp_stage_sweeps_per_s",
            "value": value,
            "unit": "stage-sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"DP (s,v,a) planner, {p.s_steps}x{p.v_steps}x{p.a_steps} grid, "
                            f"{p.t_steps} stages, straight 200 m path, 5 moving obstacles; "
                            f"one step = dist map + {sweeps_per_step} backward sweeps + forward rollout",
                "grid": [p.s_steps, p.v_steps, p.a_steps], "t_steps": p.t_steps,
                "plans_per_step_per_gpu": 1,
                "parallelism": f"independent plans x{world}" + (", all-gather of the best plan per step" if world > 1 else ""),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": ("k_backward2v" if dp.info().get("vpair") else "k_backward2") if dp.info().get("cells") == 2
                          else "k_backward",
                "launch_us": per_launch_s * 1e6,
                "algorithmic_bytes_per_launch": cells * BYTES_PER_CELL,
            },
        }
        # L2<->fabric bytes per sweep launch from the committed PMC pass of this kernel (rocprofv3
        # cannot run inside the timed process; see profiles/README.md for the command)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_dp_traffic.json")) as fd:
                tr = json.load(fd).get(f"{p.s_steps}x{p.v_steps}x{p.a_steps}")
            if tr:
                out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/r01_dp_traffic.json (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes)"
                # what actually binds the sweep: VALU issue.  Instructions per wave from the same PMC
                # passes (SQ_INSTS_VALU / SQ_WAVES), 4 cycles per wave instruction, 1024 SIMDs, against the
                # measured launch time at the 2.4 GHz peak clock (the real clock under this load is lower,
                # so the true fraction is higher).  Informative; `frac` above stays the prescribed HBM one.
                if out["roofline"]["kernel"] in ("k_backward2v", "k_backward2") and tr.get("valu_insts_per_wave"):
                    waves = cells / 128.0
                    busy = tr["valu_insts_per_wave"] * waves * 4.0 / 1024.0
                    out["roofline"]["valu_issue"] = {
                        "insts_per_wave": tr["valu_insts_per_wave"], "waves_per_launch": waves,
                        "busy_cycles_per_simd": busy, "launch_cycles_at_2.4GHz": per_launch_s * 2.4e9,
                        "frac": busy / (per_launch_s * 2.4e9)}
        except OSError:
            pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = dp_cpu_baseline(p, ep, path, occ)
        if not args.no_secondary and world == 1:
            from tpl_amd import latency
            from tpl_amd.optim import bench_ilqr
            out["secondary"] = {"batched_ilqr": bench_ilqr.run(local_rank),
                                "plan_latency": latency.run(local_rank)}
            out["secondary"]["plan_latency"]["mpc_closed_loop"]["cpu_baseline"] = mpc_cpu_baseline()
            out["secondary"]["plan_latency"]["env_update"]["cpu_baseline"] = env_cpu_baseline()
            out["secondary"]["plan_latency"]["poly_lat_update"]["cpu_baseline"] = poly_lat_cpu_baseline()
            out["secondary"]["plan_latency"]["lat_lon_update"]["cpu_baseline"] = lat_lon_cpu_baseline()
            out["secondary"]["batched_ilqr"].pop("best_record", None)

    # N > 1: the batched iLQR shards the same way -- every rank solves its own block of 1024 initial
    # states (seed = rank), the timed region is barrier-bracketed and the MAX over ranks, and the only
    # exchange is the all-gather of each rank's best {cost, id, x, u} record.
    if not args.no_secondary and world > 1:
        from tpl_amd.optim import bench_ilqr
        res = bench_ilqr.run(local_rank, cpu_budget_s=0, seed=rank, barrier=barrier, id_offset=rank * 1024)
        dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
        t = torch.tensor([res["elapsed_s"]], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        best = sharding.allgather_best(res.pop("best_record"), dist, dev)
        if rank == 0:
            el = float(t.item())
            res["value"] = world * res["updates"] * res["iterations_per_update"] / el
            res["problem_iterations_per_s"] = res["value"] * res["config"]["batch"]
            res["ms_per_update"] = el / res["updates"] * 1e3
            res["n_gpus"] = world
            res["config"]["parallelism"] = f"{world} x 1024 independent problems, all-gather of the best plan"
            res["best_plan"] = {"cost": float(best[1]), "id": int(best[2])}
            out["secondary"] = {"batched_ilqr": res}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
