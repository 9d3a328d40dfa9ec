#!/usr/bin/env python3
"""bench.py -- measures BASELINE.json's metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one batch of PLANS_PER_STEP complete DP plans on device-resident inputs (each: distance map,
t_steps-1 backward stage sweeps over the (s,v,a) grid, one forward rollout), dealt to PLANS_IN_FLIGHT planners with a
stream each (--in-flight 1: one planner, every plan behind the previous one).  value = stage sweeps per second
over the whole job (all ranks), timed over exactly K steps between barrier + device synchronize on both sides,
max over ranks.  For N > 1 the plans (independent scenarios) are sharded across ranks with no data-path
collective ("scaling": "weak"); the only exchange is ONE all-gather of the best-plan record at the end of the
timed region.

Launching: under `torch.distributed.run` (RANK / WORLD_SIZE in the environment) this process is one rank.  A
plain `python bench.py --gpus N` with N > 1 starts the N ranks ITSELF -- child processes spawned before this
process touches a GPU -- and relays rank 0's JSON line; when the box has fewer than N GPUs the ranks share
device 0 and talk over gloo ("rehearsal": true in the line: it exercises the N > 1 code path, not scaling).

Also on the line: "roofline" (HBM, algorithmic 32 B/cell/sweep over the sweep kernel's HIP-event time),
"cpu_baseline" (the CPU oracle, built -O3 -march=native on this host, timed on a bounded sample on rank 0),
and "secondary": the 8192-rollout Monte-Carlo workload of BASELINE configs[4] (1024 scenarios per GPU,
device-side reduction, one all-gather), batched iLQR iterations/s with its roofline block, plan latencies.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_MEASURED_GBS = 4910.0  # tools/micro/chip_peaks.hip on the box: copy (read + write) of 2 GiB arrays, profiles/r03_micro_chip_peaks.txt
BYTES_PER_CELL = 32            # read next float4 + write this float4
PLANS_PER_STEP = 24            # one step = one batch of plans (keeps the timed region of 20 steps above a second)
PLANS_IN_FLIGHT = 2            # planners (own HIP stream each) the batch is dealt to: the sweeps of ONE plan depend on each other,
                               # those of two plans do not -- the second plan's workgroups take the slots a sweep leaves free
                               # while its last workgroups drain (tools/two_plans_in_flight.py: 1 -> 2 streams +14 %, 3 lose)
MC_PER_GPU = 1024              # BASELINE configs[4]: 8192 Monte-Carlo rollouts on 8 GPUs
MC_CHUNK = 64                  # scenarios swept per launch (value tables for that many: 64 x 157 MB)
MC_BLOCK = 128                 # scenarios built on the host and handed to the device at a time (160 MB)


# ------------------------------------------------------------------------------------------------
# CPU baselines: the oracle (test infrastructure) timed as a baseline, never the product.  The DP and iLQR
# restatements are rebuilt here, on the host that is timed, with -O3 -march=native (oracle/Makefile, `_native/`):
# same source and results as the checker the tests use, compiler flags for speed.
# ------------------------------------------------------------------------------------------------

def _native_oracle(target):
    """Build oracle/_native/<target> on this host; returns its path or None (then the -O2 checker is timed)."""
    try:
        subprocess.check_call(["make", "-B", "-C", os.path.join(ROOT, "oracle"), f"_native/{target}"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return os.path.join(ROOT, "oracle", "_native", target)
    except (subprocess.CalledProcessError, OSError):
        return None


def dp_cpu_baseline(p, ep, path, occ, budget_s=12.0):
    """Whole stage sweeps of the same grid on all host cores until ~budget_s of wall time is spent."""
    import ctypes as C
    import numpy as np
    import oracle_dp as O
    native = _native_oracle("libdp_lon_oracle.so")
    O.use(native)
    try:
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
    finally:
        O.use(None)
    return {"value": n / el, "unit": "stage-sweeps/s", "cores": cores, "kind": "port",
            "sample": f"{n} full stage sweeps of the same {cp.s_steps}x{cp.v_steps}x{cp.a_steps} grid "
                      f"(oracle/dp_lon_oracle.c, OpenMP over rows, "
                      f"{'-O3 -march=native build of this host' if native else '-O2 checker build'})"}


def ilqr_cpu_baseline(x0, horizon, budget_s, workload="mpc"):
    """The CPU oracle on a bounded sample of the same problems, all host cores (OpenMP over problems); reset +
    solve are timed inside the C library."""
    import ctypes as C
    import numpy as np
    import oracle_ilqr as OI
    from benchkit import ilqr as bench_ilqr
    problem = bench_ilqr.WORKLOADS[workload]["problem"]
    native = _native_oracle(f"libilqr_oracle_{problem}.so")
    lib = OI.load(problem, path=native) if native else OI.load(problem)
    cores = int(lib.ilqo_max_threads())
    n = len(x0)
    opts = []
    for b in range(n):
        r = OI.OracleOptim(problem, t_cap=horizon, lib=lib)
        r.horizon = horizon
        if workload == "mpc":
            bench_ilqr._configure(r, x0[b])
        else:
            bench_ilqr._configure_lateral(r, x0[b], horizon)
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
    # and on ONE core (SURVEY 8d asks for both): the first 32 problems, one thread
    m = min(32, n)
    hs1 = (C.c_void_p * m)(*[r._h for r in opts[:m]])
    t0 = time.perf_counter()
    lib.ilqo_reset_and_update_many(hs1, m, px, 1, 1)
    el1 = time.perf_counter() - t0
    return {"value": reps * iters / el, "unit": f"batched iLQR iterations/s ({n}-problem batch)",
            "problem_iterations_per_s": reps * n * iters / el, "cores": cores, "kind": "port",
            "one_core_problem_iterations_per_s": m * iters / el1,
            "sample": f"{reps} x the same {n} problems x {iters} iterations (oracle/ilqr_oracle.cpp, OpenMP over "
                      f"problems, {'-O3 -march=native build of this host' if native else '-O2 checker build'})"}


def env_cpu_baseline(reps=20):
    """oracle/dp_env_oracle.c (one thread) on the scene of latency.env_update."""
    import numpy as np
    import oracle_env as oe
    from benchkit import scenarios
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
    from benchkit import scenarios
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
    from benchkit import scenarios
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


def ilqr_crossover(device):
    """Single-problem callers (tpl's default planner and controller): update() wall time of the HIP solver against the
    batch size next to ONE core of the CPU restatement on the same problem and settings, and the batch from which the
    HIP solver wins (SURVEY 8b callers)."""
    import numpy as np
    import oracle_ilqr as OI
    from benchkit import ilqr as bench_ilqr
    out = {}
    for workload in ("mpc", "lateral"):
        problem = bench_ilqr.WORKLOADS[workload]["problem"]
        native = _native_oracle(f"libilqr_oracle_{problem}.so")
        ref = OI.OracleOptim(problem, lib=OI.load(problem, path=native) if native else None)
        x0 = (bench_ilqr.initial_states(1) if workload == "mpc" else bench_ilqr.initial_states_lateral(1))[0]
        lat = []
        for r in range(33):
            bench_ilqr.configure_crossover(ref, workload, x0)
            t0 = time.perf_counter()
            ref.update()
            lat.append((time.perf_counter() - t0) * 1e3)
        cpu_ms = float(np.percentile(lat[3:], 50))
        rows = bench_ilqr.crossover_gpu(device, workload)
        cross = next((r["batch"] for r in rows if r["ms_per_problem"] < cpu_ms), None)
        out[workload] = {"problem": problem, "horizon": bench_ilqr.CROSSOVER[workload]["horizon"],
                         "iterations_per_update": bench_ilqr.CROSSOVER[workload]["iterations"],
                         "cpu_one_core_ms_per_problem": cpu_ms, "cpu_kind": "port (oracle/ilqr_oracle.cpp, 1 thread)",
                         "gpu": rows, "crossover_batch": cross,
                         "reading": "below crossover_batch problems per call a caller is faster on the CPU solver"}
    return out


def mpc_cpu_baseline(ticks=1000, horizon=20):
    """oracle/ilqr_oracle.cpp (one thread, one problem) in the closed loop of latency.mpc_closed_loop
    (benchkit/minisim.py): what the reference's CPU genopt solver costs per controller tick on this host."""
    import numpy as np
    import oracle_ilqr as OI
    from benchkit import minisim
    native = _native_oracle("libilqr_oracle_trajectory_tracking_mpc.so")
    opt = OI.OracleOptim("trajectory_tracking_mpc", lib=OI.load("trajectory_tracking_mpc", path=native) if native else None)
    opt.horizon = horizon
    _, trace, con = minisim.closed_loop(opt, ticks=ticks)
    lat = np.array(con.solve_ms[10:])
    return {"value": float(np.percentile(lat, 50)), "unit": "ms per update (p50)", "cores": 1, "kind": "port",
            "p99_ms": float(np.percentile(lat, 99)), "final_lateral_error_m": float(abs(trace[-1, 1])),
            "sample": f"{len(lat)} ticks of the same closed loop"}


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------

def rank_scenario(rank):
    """The headline workload of rank r: rank 0 plans the scenario the N = 1 line is quoted on; every further rank gets
    its own start speed and its own obstacle train (moved along the road and re-timed), so that the all-gathered argmin
    of an N-rank run is over N DIFFERENT plans (same grid, same cost per sweep)."""
    from benchkit import scenarios as sc
    if rank == 0:
        return sc.OBSTACLES_OVERTAKE_TRAFFIC, sc.initial_state(0.0, 8.0, 0.0)
    obst = tuple((s0 + 7.0 * rank, v + 0.25 * rank, length) for (s0, v, length) in sc.OBSTACLES_OVERTAKE_TRAFFIC)
    return obst, sc.initial_state(0.0, 8.0 + 0.5 * rank, 0.0)


def scenario_deck(rank, ep):
    """The inputs the timed region deals to its plans: FOUR obstacle sets -- the rank's headline
    scenario (rank_scenario), the reference's overtake scene 25 m nearer and 1.5 m/s faster (without the fifth car, which
    never enters the 200 m map anyway), and two seeded stress sets
    (K = 8, s0 ~ U(30, 180), v ~ U(0, 10)) -- and SEVEN start states (the headline's (0, 8, 0) and six of
    SURVEY 8d's batch variants v0 ~ U(2, 14), a0 ~ U(-1, 1)): 28 different (obstacles, start) pairs, so the data-dependent
    arms of the sweep (ACT_SINGLE_A thirds, shared / not-shared rows, the obstacle rows of pre_sdist) and of the rollout
    are not the same in every plan.  Returns (occupancy maps, start states)."""
    from benchkit import scenarios as sc
    obst0, init0 = rank_scenario(rank)
    shift = 7.0 * rank
    sets = [obst0,
            tuple((s0 - 25.0 + shift, v + 1.5, ln) for (s0, v, ln) in sc.OBSTACLES_OVERTAKE),
            sc.random_obstacles(4 + rank, 8), sc.random_obstacles(9 + rank, 6)]
    return [sc.occ_map(ep, o) for o in sets], [init0] + sc.random_initial_states(6, seed=5 + rank)


def deal(i, in_flight):
    """Which (obstacle set, start state) plan i of the timed region gets -- deterministic by plan index; the two planners in
    flight walk the deck with different offsets, the set and the state with coprime periods (4 and 7)."""
    lane, j = i % in_flight, i // in_flight
    return (j + lane) % 4, (j + 3 * lane) % 7


def dp_setup(grid, device, rank=0):
    import tpl_amd
    from benchkit import scenarios as sc
    S, V, A, T = sc.GRIDS[grid]
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    obstacles, init = rank_scenario(rank)
    occ = sc.occ_map(ep, obstacles)
    env = tpl_amd.DynProgEnvironment(device)
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    dp = tpl_amd.DynProgLonPlanner(device)
    dp.reinit_buffers(p)
    if os.environ.get("TPL_BENCH_R1") == "0":       # A/B legs only (tools/, profiles/): rule R1 = 0, one rounding per operation
        dp.set_contraction(False)
    dp.upload([init], path)
    return dp, env, p, ep, path, occ


def sweep_kernel_name(info):
    return info["sweep"]


def arithmetic_name(info):
    """The HP-1 arithmetic rule of a planner as one string (tplhip_dp_lon_info -> "rule")."""
    r = info["rule"]
    return (f"R1={r['fma']} ({'contracted like nvcc -fmad=true' if r['fma'] else 'one rounding per operation'}),"
            f"R2={('none', 'rint', 'floor')[r['quant']]},R3={('fma', 'mul-add', 'p+w(q-p)')[r['lerp']]}/{('avs', 'sva', 'sum8_f32', 'sum8_wide')[r['order']]}")


def mc_scenarios(ep, first, n):
    """Monte-Carlo scenarios [first, first + n) of the 8192-rollout workload: perturbed initial state and an own
    set of moving obstacles per scenario id (seeded by the id, so every rank count gives the same scenarios).
    The occupancy maps (31 x 201 x 50 floats per scenario: 1.25 GB per 1024) are written by ONE broadcast per call, not
    by a Python loop per scenario, slice and obstacle; mc_rollouts asks for MC_BLOCK scenarios at a time and hands each
    block to the device before it builds the next -- eight ranks set up on one host together.
    Same float64 interval arithmetic as benchkit.scenarios.occ_map (checked in tests/test_sharding.py)."""
    import numpy as np
    from benchkit import scenarios as sc
    KMAX = 6
    s0 = np.full((n, KMAX), np.nan)
    v = np.zeros((n, KMAX))
    inits = []
    for i in range(n):
        rng = np.random.default_rng(77000 + first + i)
        k = int(rng.integers(2, 7))
        s0[i, :k] = rng.uniform(30.0, 180.0, k)
        v[i, :k] = rng.uniform(0.0, 10.0, k)
        inits.append(sc.initial_state(0.0, float(rng.uniform(2.0, 14.0)), float(rng.uniform(-1.0, 1.0))))
    length, sweep_length, half_width = 4.0, 0.5, 1.5
    st = ep.s_min + np.arange(ep.s_steps) * ((ep.s_max - ep.s_min) / (ep.s_steps - 1))
    lt = ep.l_min + np.arange(ep.l_steps) * ((ep.l_max - ep.l_min) / (ep.l_steps - 1))
    lmask = np.abs(lt) <= half_width
    t = np.array([0.0 if i == 0 else ep.dt_start + (i - 1) * ep.dt for i in range(ep.t_steps)])
    lmask_f = lmask.astype(np.float32)[None, :, None, None]
    lo = s0[:, :, None] + v[:, :, None] * t[None, None, :] - 0.5 * length                                # [n, K, T]
    hi = s0[:, :, None] + v[:, :, None] * t[None, None, :] + 0.5 * length + max(sweep_length, ep.dt) * v[:, :, None]
    with np.errstate(invalid="ignore"):                                                                   # unused obstacle slots are NaN: never inside
        inside = (st[None, None, :, None] >= lo[:, :, None, :]) & (st[None, None, :, None] <= hi[:, :, None, :])     # [n, K, S, T]
    smask = inside.any(axis=1).astype(np.float32)                                                         # [n, S, T]
    occs = lmask_f * smask[:, None, :, :]
    return occs, inits


def dp_general_sweep_price(device, grid="target", plans=3):
    """What leaving the plain-fp32 sweeps costs, and how often it happens ("fast-path fences"):
    * us per sweep of this grid on the general kernel (TPLHIP_DEBUG=no_v3,no_s3: k_backward, two-pass argmin, IEEE
      divisions, any input) next to the kernel the planner selects by itself, same plan, same box;
    * the plain-fp32 sweep with both divisions as IEEE divisions (no_fastdiv) -- what a divisor without a proven
      reciprocal form pays now that it no longer changes kernels; with a path step that is not a power of two; under
      one rounding per operation (set_contraction(False): R1 = 0, the default until round 5 -- built with every division
      form like the default) and with truncated filter weights (set_filter_rule(2, 0)) -- what a non-default arithmetic
      rule costs on the same kernel; under another blend order / the eight-corner sums
      (set_blend_order(1 | 2 | 3)) -- what following such a dump of the reference build costs;
    * the kernel and the division forms the planner selects for path_step_size x s_max combinations (forms: 4 one
      multiplication, 3 two-float reciprocal, 5 reciprocal + Newton step, 0 IEEE division; dp_sweep.hip)."""
    import tpl_amd
    from benchkit import scenarios as sc
    S, V, A, T = sc.GRIDS[grid]
    T = min(T, 6)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    occ = sc.occ_map(ep, sc.OBSTACLES_OVERTAKE_TRAFFIC)
    env = tpl_amd.DynProgEnvironment(device)
    env.reinit_buffers(ep)
    env.set_occ_map(occ)
    out = {"grid": [S, V, A], "t_steps": T}
    saved = os.environ.get("TPLHIP_DEBUG")
    try:
        for key, dbg in (("selected", ""), ("path_step_0.7", ""), ("unfused", ""), ("truncated_weights", ""), ("ieee_divisions", "no_fastdiv"),
                         ("general", "no_v3,no_s3"), ("order_sva", ""), ("sum8_f32", ""), ("sum8_wide", "")):
            os.environ["TPLHIP_DEBUG"] = dbg
            p = sc.lon_params(S, V, A, T)
            if key == "path_step_0.7":                     # not a power of two: two-float reciprocal behind the tiny-dividend test
                p.path_step_size, p.path_steps = 0.7, 300
            dp = tpl_amd.DynProgLonPlanner(device)
            dp.reinit_buffers(p)
            if key == "unfused":                           # rule R1 = 0 (a template variant of the same kernel, same division forms)
                dp.set_contraction(False)
            if key == "truncated_weights":                 # rule R2 = floor
                dp.set_filter_rule(2, 0)
            if key in ("order_sva", "sum8_f32", "sum8_wide"):   # rule R3's order / form (set_blend_order): eight taps per cell-action
                dp.set_blend_order({"order_sva": 1, "sum8_f32": 2, "sum8_wide": 3}[key])
            dp.upload([sc.initial_state(0.0, 8.0, 0.0)], sc.straight_path(p.path_steps, p.path_step_size))
            dp.plan_resident(env)
            dp.sync()
            dp.timing_enable(True)
            for _ in range(plans):
                dp.plan_resident(env)
            dp.sync()
            ms, n = dp.timing_get()
            dp.timing_enable(False)
            info = dp.info()
            out[key] = {"kernel": info["sweep"], "us_per_sweep": ms / max(1, n) * 1e3, "division_forms": [info["path_div3"], info["range_div3"]]}
            del dp
        os.environ["TPLHIP_DEBUG"] = ""
        table = []
        for s_max in (100.0, 157.3, 200.0):
            for step in (0.1, 0.2, 0.25, 0.3, 0.5, 0.7, 0.75, 1.0, 1.3, 2.0):
                p = sc.lon_params(S, V, A, 3)
                p.s_max = s_max
                p.path_step_size = step
                p.path_steps = int(s_max / step) + 8
                dp = tpl_amd.DynProgLonPlanner(device)
                dp.reinit_buffers(p)
                ep3 = sc.env_params(t_steps=3, s_steps=201, l_steps=31)
                ep3.s_max = s_max
                env3 = tpl_amd.DynProgEnvironment(device)
                env3.reinit_buffers(ep3)
                env3.set_occ_map(sc.occ_map(ep3, sc.OBSTACLES_OVERTAKE_TRAFFIC))
                dp.update(sc.initial_state(0.0, 8.0, 0.0), sc.straight_path(p.path_steps, p.path_step_size), env3)
                info = dp.info()
                del env3
                table.append({"path_step_size": step, "s_max": s_max, "kernel": info["sweep"],
                              "division_forms": [info["path_div3"], info["range_div3"]]})
                del dp
        out["selection"] = table
        out["on_plain_fp32_sweeps"] = sum(1 for r in table if r["kernel"] != "k_backward") / len(table)
    finally:
        if saved is None:
            os.environ.pop("TPLHIP_DEBUG", None)
        else:
            os.environ["TPLHIP_DEBUG"] = saved
    return out


class _DeviceBytes:
    """Zero-copy view of device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def mc_rollouts(device, rank, world, dist, rehearsal, barrier, steps=2, warmup_chunks=2, grid="config2",
                per_gpu=MC_PER_GPU, chunk=MC_CHUNK):
    """BASELINE configs[4]: 1024 scenario rollouts per GPU on the 128x64x32x50 grid, swept MC_CHUNK scenarios per
    launch; per-rank lexicographic (constr, cost, id) reduction on the device; ONE all-gather of the 1416-byte
    record straight from HBM; identical argmin on every rank.  (Also run on the reference's default grid, the size
    tpl really plans on: `grid`, `per_gpu`, `chunk`.)"""
    import numpy as np
    import torch
    import tpl_amd
    from benchkit import scenarios as sc
    from tpl_amd import sharding
    MC_PER_GPU, MC_CHUNK = per_gpu, chunk
    S, V, A, T = sc.GRIDS[grid]
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    n_total = MC_PER_GPU * world
    lo, hi = sharding.shard_range(n_total, rank, world)
    n = hi - lo
    t_setup0 = time.perf_counter()
    dp = tpl_amd.DynProgLonPlanner(device)
    dp.reinit_buffers(p)
    # the scenario set goes up block by block: one 128-scenario buffer on the host (160 MB), not the whole 1.25 GB
    dp.upload_scenarios(None, ep, None, path, chunk=MC_CHUNK, n=n)
    for b0 in range(0, n, MC_BLOCK):
        occs, inits = mc_scenarios(ep, lo + b0, min(MC_BLOCK, n - b0))
        dp.set_scenario_block(b0, occs, inits)
    del occs
    setup_s = time.perf_counter() - t_setup0

    def one_pass():
        for first in range(0, n, MC_CHUNK):
            dp.plan_scenarios(first, min(MC_CHUNK, n - first), id_base=lo, reset_best=(first == 0))

    for c in range(warmup_chunks):
        dp.plan_scenarios(c * MC_CHUNK, min(MC_CHUNK, n - c * MC_CHUNK), id_base=lo, reset_best=True)
    dp.sync()
    dp.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    # the one exchange of the path: every rank's record -> all ranks, sent from device memory
    ptr, nbytes = dp.best_record_device()                    # synchronises this rank's stream
    if dist is not None:
        # (a copy inside HBM into memory of torch's own allocator: RCCL then never sees a foreign hipMalloc'd range)
        mine = torch.as_tensor(_DeviceBytes(ptr, nbytes), device=torch.device("cuda", device)).clone()
        if rehearsal:
            mine = mine.cpu()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        recs = [g.cpu().numpy() for g in gathered]
    else:
        recs = [dp.best_record_bytes()]
    barrier()
    el = time.perf_counter() - t0
    sweep_ms, launches = dp.timing_get()
    dp.timing_enable(False)

    def key(r):
        return (float(r[0:4].view(np.float32)[0]), float(r[4:8].view(np.float32)[0]), int(r[8:12].view(np.int32)[0]))
    best = min(recs, key=key)
    bk = key(best)
    identical = True
    if dist is not None:
        ids = torch.tensor([bk[2], -bk[2]], dtype=torch.int64, device="cpu" if rehearsal else torch.device("cuda", device))
        dist.all_reduce(ids, op=dist.ReduceOp.MAX)
        identical = int(ids[0].item()) == bk[2] and int(-ids[1].item()) == bk[2]
        t = torch.tensor([el, setup_s], dtype=torch.float64, device="cpu" if rehearsal else torch.device("cuda", device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, setup_s = float(t[0].item()), float(t[1].item())            # both: the slowest rank's
    cells = S * V * A
    per_launch_s = (sweep_ms / 1e3) / max(1, launches)
    achieved = MC_CHUNK * cells * BYTES_PER_CELL / per_launch_s / 1e9
    res = {
        "metric": "dp_stage_sweeps_per_s", "unit": "stage-sweeps/s (one sweep = one stage of one scenario)",
        "value": steps * n_total * (T - 1) / el, "n_gpus": world, "scaling": "weak",
        "rollouts": n_total, "rollouts_per_gpu": MC_PER_GPU, "rollouts_per_s": steps * n_total / el,
        "plans_per_s": steps * n_total / el,
        "ms_per_pass": el / steps * 1e3, "passes": steps,
        "config": {"workload": f"{n_total} Monte-Carlo scenario rollouts (perturbed initial state + own moving obstacles), "
                               f"DP grid {S}x{V}x{A}, {T} stages, {MC_CHUNK} scenarios per launch, device-side "
                               "(constr, cost, id) reduction, one all-gather of the best-plan record",
                   "grid": [S, V, A], "t_steps": T},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": dp.info().get("scenario_kernel", "k_backward"),
                     "launch_us": per_launch_s * 1e6, "scenarios_per_launch": MC_CHUNK,
                     "algorithmic_bytes_per_launch": MC_CHUNK * cells * BYTES_PER_CELL, "rank": rank},
        "best_plan": {"constr": bk[0], "cost": bk[1], "id": bk[2], "identical_on_all_ranks": bool(identical)},
        "record_bytes": int(nbytes),
        "setup_s": setup_s,
    }
    traffic_block(res["roofline"], f"{S}x{V}x{A}@{MC_CHUNK}")
    return res


def traffic_block(roofline, key):
    """L2<->fabric bytes per sweep launch and the issue-model fractions from the committed PMC passes of this kernel
    (rocprofv3 cannot run inside the timed process; profiles/README.md has the commands).  `key`: "SxVxA" or "SxVxA@N"."""
    for f in ("r06_dp_traffic.json", "r05_dp_traffic.json", "r04_dp_traffic.json", "r03_dp_traffic.json", "r02_dp_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", f)) as fd:
                tr = json.load(fd).get(key)
        except OSError:
            continue
        if tr and tr.get("kernel", roofline["kernel"]) == roofline["kernel"]:
            roofline["traffic"] = tr["traffic_bytes_per_launch"]
            roofline["traffic_source"] = (f"profiles/{f} (2*FETCH_SIZE + WRITE_SIZE, factor calibrated for 12-byte lanes, separate --pmc passes); "
                                          "a constant of the repository, not measured in this run: rocprofv3 cannot run inside the timed process")
            roofline["traffic_profiled_kernel_rev"] = tr.get("profiled_kernel_rev", "before round 4")
            for k in ("valu_issue", "texture_addr", "l2_hit_rate"):
                if k in tr:
                    roofline[k] = tr[k]
            if "valu_issue" in tr:
                # what binds, as the PMC passes say it: `frac` stays the HBM fraction SURVEY 8d prescribes
                # (algorithmic bytes per launch / launch time / 8 TB/s); `bound` names the resource the kernel is up against
                roofline["bound"] = "valu_issue"
                roofline["prescribed_bound"] = "hbm"
                roofline["binding"] = (f"not HBM: VALU issue {tr['valu_issue']:.2f} and the texture-address path {tr.get('texture_addr', 0.0):.2f} of the "
                                       f"kernel's cycles (PMC instruction counts x measured per-instruction costs, model in profiles/{f}); neither "
                                       "binds alone -- the rest is the latency of one row round trip per action at eight waves per SIMD (docs/NOTEBOOK.md 2.2b')")
            return


def worker(args):
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line, the result.  Libraries write there too (RCCL prints its version banner through C stdio,
    # gloo announces connections): from here on file descriptor 1 is stderr, and the line goes to the saved descriptor
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    # Rehearsal: the N > 1 code path on a box with fewer GPUs than ranks -- every rank on device 0, collectives over
    # gloo (RCCL refuses two ranks on one device).  Set by the launcher below, never by the driver's torchrun.
    rehearsal = world > 1 and os.environ.get("TPL_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # TPL_BENCH_FORCE_DIST=1 (tests/test_bench_gpu.py): a world of ONE rank still goes through RCCL -- process group,
    # barrier, the device-memory all-gather of the record -- so that the N > 1 code path runs on the real backend
    # on a one-GPU box
    if world > 1 or os.environ.get("TPL_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if not rehearsal and torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
            raise SystemExit(f"bench.py --gpus {world}: this node shows {torch.cuda.device_count()} GPU(s) to rank {rank}; "
                             f"one process per GPU needs {world} (lower --gpus, or widen HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    torch.cuda.set_device(local_rank)
    collective = None
    if dist is not None:
        # what the driver cannot see from n_gpus = WORLD_SIZE alone: how many ranks the process group really holds (an
        # all-reduce of ones through the backend) and that every rank sits on a device of its own
        assert dist.get_world_size() == world and dist.get_rank() == rank, (dist.get_world_size(), world, dist.get_rank(), rank)
        if not rehearsal:
            assert local_rank < torch.cuda.device_count(), f"LOCAL_RANK {local_rank} but {torch.cuda.device_count()} devices"
        cdev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
        ones = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        devs = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([local_rank], dtype=torch.int64, device=cdev))
        collective = {"backend": dist.get_backend(), "ranks_seen": int(ones.item()), "devices": [int(d.item()) for d in devs]}
        assert collective["ranks_seen"] == world, collective

    from tpl_amd import sharding
    t_setup0 = time.perf_counter()
    dp, env, p, ep, path, occ = dp_setup(args.grid, local_rank, rank)
    cells = p.s_steps * p.v_steps * p.a_steps
    sweeps_per_plan = p.t_steps - 1
    plans_per_step = 2 if args.grid == "config4" else PLANS_PER_STEP

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the batch of a step is dealt to PLANS_IN_FLIGHT planners, each with its own stream and tables, and with its own
    # environments (an environment holds the distance map a plan writes) -- one per obstacle set of the deck.  Every plan
    # gets its own (obstacle set, start state) pair, deterministic by plan index (scenario_deck / deal): upload() of the
    # start state + plan_resident() on that set's environment, nothing waited for in between.
    import tpl_amd
    in_flight = args.in_flight if args.in_flight is not None else (1 if args.grid == "config4" else PLANS_IN_FLIGHT)
    in_flight = max(1, min(in_flight, plans_per_step))
    occs, starts = scenario_deck(rank, ep)
    # one plan at a time (what a tpl caller does): all stages of a plan in ONE launch; plans in flight: stage by stage
    chained = in_flight == 1 if args.chained == "auto" else args.chained == "1"
    fleet = []
    for lane in range(in_flight):
        q = dp if lane == 0 else dp_setup(args.grid, local_rank, rank)[0]
        q.set_chained_sweep(1 if chained else 0)
        envs = []
        for o in occs:
            e = tpl_amd.DynProgEnvironment(local_rank)
            e.reinit_buffers(ep)
            e.set_occ_map(o)
            envs.append(e)
        fleet.append((q, envs))
    del env
    counter = [0]                                    # plans dealt so far (warm-up included: the deck keeps turning)
    last = {}                                        # lane -> (obstacle set, start state) of its most recent plan

    def run(n_steps):
        for _ in range(n_steps * plans_per_step):
            i = counter[0]
            counter[0] += 1
            lane = i % in_flight
            q, envs = fleet[lane]
            io, ist = deal(i, in_flight)
            q.upload([starts[ist]], path)
            q.plan_resident(envs[io])
            last[lane] = (io, ist)

    def sync_all():
        for q, _ in fleet:
            q.sync()

    def plan_hash(traj):
        return sharding.plan_record(np.array([[s.t, s.s, s.v, s.a, s.j, s.cost, s.constr] for s in traj.states], dtype=np.float32), 0).tobytes()

    # warm-up, first half: every pair of the deck once on the LAST lane's planner, one at a time -- the records the spot
    # check after the timed region compares against (another planner, another moment, same bits)
    want = {}
    qa, ea = fleet[-1]
    for io in range(len(occs)):
        for ist in range(len(starts)):
            qa.upload([starts[ist]], path)
            qa.plan_resident(ea[io])
            want[(io, ist)] = plan_hash(qa.download()[0])
    assert len(set(want.values())) > len(want) // 2, "the deck's plans are (nearly) all the same: inputs are not varied"
    run(args.warmup)
    sync_all()
    # the sweep kernel ALONE on the chip (one plan at a time on one stream, six plans behind two untimed ones): the launch
    # duration the roofline's per-launch figures are quoted on -- with two plans in flight a launch shares the chip and
    # its duration says little by itself
    for k in range(2):                                  # (one plan at a time from here on: two untimed ones first)
        dp.upload([starts[k]], path)
        dp.plan_resident(fleet[0][1][k])
    dp.sync()
    dp.timing_enable(True)
    for k in range(6):
        dp.upload([starts[k]], path)
        dp.plan_resident(fleet[0][1][k % len(fleet[0][1])])
    alone_ms, alone_launches = dp.timing_get()
    dp.timing_enable(False)
    chained_ran = chained and bool(dp.info()["chained_sweep"])       # (asked for AND this grid can be chained)
    # ... the same six plans with the OTHER launch form (chained <-> stage by stage), microseconds per stage sweep:
    # `roofline.stage_us_by_launch_form`
    # (--no-form-ab: skipped, for kernel traces -- both forms run the same kernel, and a trace's average would mix them)
    form_ms, form_launches, other_form_chained = 0.0, 0, False
    if not args.no_form_ab:
        dp.set_chained_sweep(0 if chained_ran else 1)
        for k in range(2):
            dp.upload([starts[k]], path)
            dp.plan_resident(fleet[0][1][k])
        dp.sync()
        other_form_chained = bool(dp.info()["chained_sweep"])
        dp.timing_enable(True)
        for k in range(6):
            dp.upload([starts[k]], path)
            dp.plan_resident(fleet[0][1][k % len(fleet[0][1])])
        form_ms, form_launches = dp.timing_get()
        dp.timing_enable(False)
        dp.set_chained_sweep(1 if chained else 0)
    # ... and the same kernel under the OTHER setting of rule R1 (one rounding per operation; a template variant with the same
    # division forms), same box, same minute: `roofline.launch_us_by_rule`.  Rank 0 only, outside the
    # timed region, on a planner of its own that is released before timing starts.
    other_rule_us = None
    if rank == 0:
        q0 = dp_setup(args.grid, local_rank, rank)[0]
        q0.set_contraction(not bool(dp.info()["rule"]["fma"]))
        for k in range(4):                              # (both table sets of the back-to-back mode touched before timing)
            q0.upload([starts[k]], path)
            q0.plan_resident(fleet[0][1][k])
        q0.sync()
        q0.timing_enable(True)
        for k in range(6):
            q0.upload([starts[k]], path)
            q0.plan_resident(fleet[0][1][k % len(fleet[0][1])])
        ms0, n0 = q0.timing_get()
        other_rule_us = (arithmetic_name(q0.info()).split(" ")[0], ms0 / max(1, n0) * 1e3)
        del q0
    setup_s = {"headline": time.perf_counter() - t_setup0}
    for q, _ in fleet:
        q.timing_enable(True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    sync_all()
    # N > 1: the path's only exchange -- ONE all-gather of every rank's best-plan record (1.4 KB),
    # followed by the same lexicographic argmin on every rank
    traj = dp.download()[0]
    rec = sharding.plan_record(np.array([[s.t, s.s, s.v, s.a, s.j, s.cost, s.constr] for s in traj.states], dtype=np.float32), rank)
    best = sharding.allgather_best(rec, dist, torch.device("cpu") if rehearsal else torch.device("cuda", local_rank))
    barrier()
    elapsed = time.perf_counter() - t0
    rank0_elapsed = elapsed                         # this rank's own clock (the roofline block is rank 0's)
    sweep_ms = sweep_launches = 0
    for q, _ in fleet:
        ms, n = q.timing_get()
        sweep_ms, sweep_launches = sweep_ms + ms, sweep_launches + n
        q.timing_enable(False)
    # spot check: the LAST plan of every lane against the record of its (obstacle set, start state) pair from warm-up
    spot = 0
    for lane, (q, _) in enumerate(fleet):
        assert plan_hash(q.download()[0]) == want[last[lane]], f"lane {lane}: plan of deck pair {last[lane]} differs from its warm-up record"
        spot += 1

    rank_values = rank_plans = None
    if dist is not None:
        cdev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
        mine = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        every = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(every, mine)
        own_sweeps = sweeps_per_plan * plans_per_step * args.steps
        rank_values = [own_sweeps / float(e.item()) for e in every]           # each rank's own sweeps/s over ITS clock
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own plan (cost, constraint sum, speed at the last stage): N different scenarios give N different entries
        plan = torch.tensor([float(rec[1]), float(rec[0]), float(traj.states[-1].v)], dtype=torch.float64, device=cdev)
        plans = [torch.zeros(3, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(plans, plan)
        rank_plans = [{"cost": float(q[0]), "constr": float(q[1]), "v_end": float(q[2])} for q in plans]
        su = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(su, torch.tensor([setup_s["headline"]], dtype=torch.float64, device=cdev))
        setup_s["headline_per_rank"] = [float(q.item()) for q in su]

    total_sweeps = sweeps_per_plan * plans_per_step * args.steps * world
    out = None
    if rank == 0:
        # Per launch (HIP events on each planner's stream, summed): with one plan in flight that IS the kernel's duration
        # and `achieved` = bytes per launch / duration, as SURVEY 8d prescribes.  With two in flight a launch shares the
        # chip for its whole life: its duration (what rocprofv3 lists too) nearly doubles while the chip moves two launches'
        # bytes -- there `achieved` = the bytes of ALL sweep launches of the timed region / the timed region (a lower bound:
        # the region holds the fronts, rollouts and gaps too), and the per-launch figures sit beside it.
        per_launch_s = (sweep_ms / 1e3) / max(1, sweep_launches)
        alone_s = (alone_ms / 1e3) / max(1, alone_launches)
        own_launches = sweeps_per_plan * plans_per_step * args.steps
        # `achieved` / `frac`: ALWAYS the per-launch figure SURVEY 8d prescribes -- algorithmic bytes of one launch / the
        # kernel's duration alone on the chip (HIP events on the launching stream; with one plan in flight that is the
        # timed region's own average, with two it is the four-plan run on one stream just before it).  What two plans in
        # flight add is reported under `in_flight`, never mixed into `frac`.
        launch_s = per_launch_s if in_flight == 1 else alone_s
        achieved = cells * BYTES_PER_CELL / launch_s / 1e9
        aggregate = own_launches * cells * BYTES_PER_CELL / rank0_elapsed / 1e9
        out = {
            "metric": "dp_stage_sweeps_per_s",
            "value": total_sweeps / elapsed,
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
                            f"{p.t_steps} stages, straight 200 m path; one step = {plans_per_step} plans, each with its own start state "
                            f"and one of 4 obstacle sets (28 pairs, dealt by plan index): upload + dist map + {sweeps_per_plan} backward sweeps + forward rollout",
                "grid": [p.s_steps, p.v_steps, p.a_steps], "t_steps": p.t_steps,
                "plans_per_step_per_gpu": plans_per_step, "ms_per_plan": elapsed / args.steps / plans_per_step * 1e3,
                "plans_in_flight_per_gpu": in_flight,
                # all stages of a plan behind the terminal one in ONE launch (tplhip_dp_lon_set_chained_sweep): the launch form of
                # the timed region's plans -- chosen with one plan in flight, not with two
                "chained_sweep": chained_ran,
                # the HP-1 arithmetic contract the sweeps ran under (tplhip_dp_lon_info): R1 = 1 is nvcc's default
                # contraction of the reference's device code, the best estimate of what the reference binary computes
                "arithmetic": arithmetic_name(dp.info()),
                "inputs": {"obstacle_sets": len(occs), "start_states": len(starts), "plans_spot_checked": spot},
                "parallelism": f"independent plans x{world}" + (", one all-gather of the best plan" if world > 1 else ""),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": sweep_kernel_name(dp.info()),
                # the kernel's duration ALONE on the chip (HIP events; the timed region's own average with one plan in flight): what
                # a trace of `--in-flight 1` lists
                "launch_us": launch_s * 1e6,
                # the kernel alone under both settings of rule R1 (HIP events, plans on one stream just before the timed region)
                "launch_us_by_rule": {arithmetic_name(dp.info()).split(" ")[0]: alone_s * 1e6, other_rule_us[0]: other_rule_us[1]},
                # microseconds per STAGE SWEEP of one plan at a time under both launch forms (HIP events around a plan's sweeps /
                # its stages; a chained launch sweeps `stages_per_chained_launch` stages, which is what a kernel trace lists)
                "stage_us_by_launch_form": dict(
                    [(("chained" if chained_ran else "stage_by_stage"), alone_s * 1e6)]
                    + ([(("stage_by_stage" if chained_ran else ("chained" if other_form_chained else "stage_by_stage_again")),
                         form_ms / max(1, form_launches) * 1e3)] if form_launches else [])),
                "stages_per_chained_launch": max(0, p.t_steps - 2),
                "algorithmic_bytes_per_launch": cells * BYTES_PER_CELL,
                "achieved_is": ("algorithmic bytes of one stage sweep / (the HIP-event span of a plan's chained launch / its stages), alone on the chip"
                                if chained_ran else "algorithmic bytes per launch / the launch's HIP-event duration alone on the chip")
                               + ("" if in_flight == 1 else " (six plans on one stream just before the timed region)"),
                # the whole timed region: with two plans in flight launches overlap for their whole life -- the events of a stream
                # also span the waits behind the other stream's launches (`stream_us_per_launch`), and a trace lists start-to-end
                # durations of launches that share the chip (profiles/README.md); the chip's rate is the bytes of ALL the rank's
                # sweep launches / the region (a lower bound: the region holds uploads, fronts, rollouts and gaps too)
                "in_flight": {"launches_in_flight": in_flight, "achieved": aggregate, "frac": aggregate / HBM_PEAK_GBS,
                              "stream_us_per_launch": per_launch_s * 1e6, "effective_us_per_launch": rank0_elapsed / own_launches * 1e6},
                # context only (`frac` stays against the prescribed 8 TB/s): what a plain copy kernel reaches on this part
                "frac_of_measured_copy_rate": achieved / HBM_COPY_MEASURED_GBS,
                "measured_copy_rate_source": "profiles/r03_micro_chip_peaks.txt (read 6.5, copy 4.9, triad 5.4 TB/s)",
            },
        }
        if rehearsal:
            out["rehearsal"] = True
            out["note"] = f"{world} ranks share ONE device over gloo: exercises the N > 1 code path, says nothing about scaling"
        out["setup_s"] = setup_s
        if dist is not None:
            out["best_plan"] = {"constr": float(best[0]), "cost": float(best[1]), "rank": int(best[2])}
            out["collective"] = dict(collective, per_rank_value=rank_values, per_rank_plan=rank_plans)
        traffic_block(out["roofline"], f"{p.s_steps}x{p.v_steps}x{p.a_steps}")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = dp_cpu_baseline(p, ep, path, occ)
    del dp, fleet

    secondary = {}
    if not args.no_secondary:
        # BASELINE configs[4] at this job size: 1024 rollouts per GPU
        mc = mc_rollouts(local_rank, rank, world, dist, rehearsal, barrier)
        # the grid tpl really plans on (reference default 201x37x20, 10 stages), 256 scenarios per GPU, 64 per launch
        mc_default = mc_rollouts(local_rank, rank, world, dist, rehearsal, barrier, steps=4, grid="default", per_gpu=256)
        from benchkit import ilqr as bench_ilqr
        il = bench_ilqr.run(local_rank, seed=rank, barrier=barrier if dist is not None else None, id_offset=rank * 1024)
        lat = bench_ilqr.run(local_rank, seed=rank, barrier=barrier if dist is not None else None, id_offset=rank * 1024,
                             workload="lateral", updates=20)
        x0_mpc, x0_lat = il.pop("x0"), lat.pop("x0")
        if dist is not None:
            dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
            for res in (il, lat):
                t = torch.tensor([res["elapsed_s"]], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                bestp = sharding.allgather_best(res.pop("best_record"), dist, dev)
                el = float(t.item())
                res["value"] = world * res["updates"] * res["iterations_per_update"] / el
                res["problem_iterations_per_s"] = res["value"] * res["config"]["batch"]
                res["ms_per_update"] = el / res["updates"] * 1e3
                res["n_gpus"] = world
                res["config"]["parallelism"] = f"{world} x {res['config']['batch']} independent problems, one all-gather of the best plan"
                res["best_plan"] = {"cost": float(bestp[1]), "id": int(bestp[2])}
                res["roofline"]["rank"] = rank
        il.pop("best_record", None)
        lat.pop("best_record", None)
        secondary = {"mc_rollouts": mc, "default_grid_batched": mc_default, "batched_ilqr": il,
                     "batched_ilqr_lateral_profile": lat}
        if rank == 0 and world == 1:
            if not args.no_cpu_baseline:
                il["cpu_baseline"] = ilqr_cpu_baseline(x0_mpc, 50, 10.0, "mpc")
                lat["cpu_baseline"] = ilqr_cpu_baseline(x0_lat, 250, 6.0, "lateral")
            il["batch_sweep"] = bench_ilqr.batch_sweep(local_rank)
            secondary["general_sweep"] = dp_general_sweep_price(local_rank)
            if not args.no_cpu_baseline:
                secondary["ilqr_single_problem_crossover"] = ilqr_crossover(local_rank)
            from benchkit import latency
            secondary["plan_latency"] = latency.run(local_rank)
            if not args.no_cpu_baseline:
                pl = secondary["plan_latency"]
                pl["mpc_closed_loop"]["cpu_baseline"] = mpc_cpu_baseline()
                pl["env_update"]["cpu_baseline"] = env_cpu_baseline()
                pl["poly_lat_update"]["cpu_baseline"] = poly_lat_cpu_baseline()
                pl["lat_lon_update"]["cpu_baseline"] = lat_lon_cpu_baseline()
    if rank == 0:
        if secondary:
            out["secondary"] = secondary
            # BASELINE's metric has three parts: "DP stage-sweeps/s + batched iLQR iters/s; p50 plan latency".  `value` is the
            # first; the other two are mirrored from `secondary` into objects the driver's record keeps
            il = secondary["batched_ilqr"]
            ilr = il["roofline"]
            out["roofline"]["ilqr"] = {
                "value": il["value"], "unit": il["unit"], "config": il["config"]["workload"], "dtype": il["dtype"],
                "problem_iterations_per_s": il["problem_iterations_per_s"],
                "bound": ilr["bound"], "achieved": ilr["achieved"], "peak": ilr["peak"], "roofline_unit": ilr["unit"], "frac": ilr["frac"],
                "kernel": ilr["kernel"], "ms_per_update": il["ms_per_update"], "kernel_ms_per_update": il["kernel_ms_per_update"],
                "update_call_ms": il.get("update_call_ms"), "n_gpus": il.get("n_gpus", 1),
            }
            if "cpu_baseline" in il:
                out["roofline"]["ilqr"]["cpu_baseline"] = il["cpu_baseline"]
            if "plan_latency" in secondary:
                pl = secondary["plan_latency"]
                lat = {"dp_default_grid": pl["dp_default_plan"]["p50_ms"],
                       # update() on the headline grid: the call's default (chained sweep) and the same call stage by stage
                       "dp_target_grid": pl["dp_target_plan"]["p50_ms"],
                       "dp_target_grid_stage_by_stage": pl["dp_target_plan"]["p50_ms_stage_by_stage"],
                       "mpc_tick": pl["mpc_closed_loop"]["p50_ms"],
                       "env_update": pl["env_update"]["p50_ms"], "poly_lat_update": pl["poly_lat_update"]["p50_ms"],
                       "lat_lon_update": pl["lat_lon_update"]["p50_ms"],
                       "what": "p50 wall time of one call through the drop-in surface, host buffers in and out (benchkit/latency.py); "
                               "dp_default_grid = DynProgLonPlanner.update on the reference's default 201x37x20x10 grid, "
                               "mpc_tick = Optim.update() of the tracking MPC (N = 20) in the closed loop of configs[0]"}
                cb = pl["mpc_closed_loop"].get("cpu_baseline")
                if cb:
                    lat["mpc_tick_cpu_baseline"] = cb.get("p50_ms", cb.get("value"))
                out["config"]["p50_plan_latency_ms"] = lat
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + "\n").encode())


def launcher(args, argv):
    """`python bench.py --gpus N` without torchrun: start the N ranks as CHILD processes (this process never
    initialises the GPU: device_count() does not, and nothing else here touches HIP), wait, relay rank 0's line."""
    import socket
    import torch
    n_dev = torch.cuda.device_count()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if n_dev < args.gpus:
            env["TPL_BENCH_REHEARSAL"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = 1 if procs[0].returncode != 0 else 0          # a rank killed by a signal has a NEGATIVE code: any non-zero is failure
    for pr in procs[1:]:
        # every rank is waited for (no short circuit); once one has failed the others may sit in a collective that
        # will never complete: give them a moment, then end them -- exactly the processes started above, by handle
        try:
            code = pr.wait(timeout=30 if rc else None)
        except subprocess.TimeoutExpired:
            pr.terminate()
            try:
                code = pr.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pr.kill()
                code = pr.wait()
        rc = rc or (1 if code != 0 else 0)
    # ONE JSON line on stdout: rank 0's (gloo announces its connections on stdout in the rehearsal set-up)
    lines = [ln for ln in out.decode("utf8", "replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    return rc if lines else max(rc, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", default="target", help="target (256x128x64, default) | config2 | config4 | default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--chained", default="auto", choices=["auto", "0", "1"],
                    help="chained sweep (all stages of a plan in one launch, DynProgLonPlanner.set_chained_sweep): auto = with "
                         "one plan in flight (+7-10 %% sweeps/s there), not with two (-4 %%); 0 / 1 force it")
    ap.add_argument("--no-form-ab", action="store_true", help="skip the six plans under the other launch form (kernel traces)")
    ap.add_argument("--in-flight", type=int, default=None,
                    help=f"planners (streams) a step's plans are dealt to; 1 = every plan behind the previous one "
                         f"(default {PLANS_IN_FLIGHT}; 1 on config4, whose sweeps are 16 residency rounds deep and gain nothing)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launcher(args, sys.argv[1:]))
    worker(args)


if __name__ == "__main__":
    main()
