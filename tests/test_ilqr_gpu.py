"""GPU parity tests of HP-2: the batched HIP iLQR solvers (through the C-ABI of
models/libilqr_<problem>.so) against the golden vectors captured from the
reference genopt solver, and against the CPU oracle.

Tolerance (fp64; the reference is built -O3 -ffast-math, so bitwise equality is
not defined): trajectories (x, u, multipliers) within 1e-6 of the array's
largest magnitude -- ten times tighter than the north star's 1e-5 --,
trajectory costs within 1e-8 relative, derivative blocks within 1e-6 / 1e-5,
iteration counts and line-search outcomes equal (tests/ilqr_cases.compare)."""
import os
import sys

import numpy as np
import pytest

import ilqr_cases as cases
import oracle_ilqr as OI

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _factory(model):
    from tpl_amd.optim import genopt, optimizers as o
    cfg = getattr(o, "config_" + model)()
    return genopt.build(cfg, model)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_hip_solver_matches_reference_golden(name):
    model, _, kw, _ = cases.CASES[name]
    g = np.load(os.path.join(GOLDEN, f"ilqr_{name}.npz"))
    opt = _factory(model)()
    cases.apply_inputs(opt, g)
    assert np.allclose(opt.dynamics(g["dyn.x"], g["dyn.u"], 3, 0.07), g["dyn.dynamics"], rtol=1e-12, atol=1e-14)
    assert np.allclose(opt.ct_dynamics(g["dyn.x"], g["dyn.u"], 3, 0.07), g["dyn.ct_dynamics"], rtol=1e-12, atol=1e-14)
    snaps = cases.run_sequence(opt, int(g["in.shift"]))
    bad = cases.compare(snaps, g, first_order=bool(kw.get("first_order")))
    assert not bad, "\n".join(bad)


def test_setup_idioms_of_the_callers_work_on_views():
    """In-place mutation of views, broadcasting setters, aliases
    (control/model_predictive_controller.py:69-76, 134-155, 192)."""
    opt = _factory("trajectory_tracking_mpc")()
    cases.mpc(opt, seed=0)
    assert opt.horizon == 50 and opt.T == 50 and opt.step == 0.05 and opt.dt == 0.05
    assert opt.x.shape == (51, 7) and opt.u.shape == (50, 2) and opt.K.shape == (50, 2, 7)
    assert np.all(opt.u_min[:, 0] == -3.0) and np.all(opt.lg_mult_limit == 0.0)
    assert opt.params.__slots__[:3] == ["pd", "pv", "pdelta"]
    opt.params.ref_v[3] = 5.0                      # callers edit array parameters in place (velocity_optim.py:269)
    assert opt.params.ref_v[3] == 5.0
    with pytest.raises(ValueError):
        opt.x = np.zeros((3, 3))
    with pytest.raises(AttributeError):
        opt.nonexistent = 1
    opt.horizon = 1000
    assert opt.horizon == 299                      # optim.c:1726-1734
    one = _factory("lateral_profile")()
    assert one.u.shape == (20,) and one.x.shape == (21, 2)      # U == 1 is squeezed (optim.c:1319-1324)


def test_batch_equals_single_problems():
    make = _factory("trajectory_tracking_mpc")
    B = 5
    batch = make(batch=B, horizon=50)
    singles = []
    for b in range(B):
        one = make()
        cases.mpc(one, seed=10 + b)
        singles.append(one)
    cases.mpc(batch, seed=0)                       # common settings + parameters ...
    for b in range(B):                             # ... per-problem initial states
        batch.x[b, 0] = singles[b].x[0]
    batch.update()
    for b in range(B):
        singles[b].update()
        assert np.array_equal(batch.x[b], singles[b].x)
        assert np.array_equal(batch.u[b], singles[b].u)
        assert batch.traj_costs[b] == singles[b].traj_costs
        assert batch.iterations[b] == singles[b].iterations


def test_bench_batch_matches_oracle():
    """BASELINE config 3: 1024 random initial states, kinematic single-track MPC,
    N = 50, 20 iterations fixed (min_rel_cost_change = 0).  Every 64th problem is
    re-solved by the CPU oracle; all costs must be finite and improved."""
    make = _factory("trajectory_tracking_mpc")
    B = 1024
    opt = make(batch=B, horizon=50)
    cases.mpc(opt, seed=0, min_rel=0.0)
    rng = np.random.default_rng(0)
    x0 = np.zeros((B, 7))
    x0[:, 1] = rng.uniform(-1, 1, B)
    x0[:, 2] = rng.uniform(-0.2, 0.2, B)
    x0[:, 4] = rng.uniform(4, 10, B)
    opt.x[:, 0] = x0
    opt.update()
    assert np.all(opt.iterations == 20) and np.all(np.isfinite(opt.traj_costs))
    for b in range(0, B, 64):
        ref = OI.OracleOptim("trajectory_tracking_mpc")
        cases.mpc(ref, seed=0, min_rel=0.0)
        ref.x[0] = x0[b]
        ref.update()
        assert cases.scaled_err(opt.x[b], ref.x) < 1e-6, b
        assert cases.scaled_err(opt.u[b], ref.u) < 1e-6, b
        assert abs(opt.traj_costs[b] - ref.traj_costs) < 1e-8 * abs(ref.traj_costs), b
    # resident stepping gives the same answer as update()
    opt2 = make(batch=B, horizon=50)
    cases.mpc(opt2, seed=0, min_rel=0.0)
    opt2.x[:, 0] = x0
    opt2.upload()
    opt2.run_resident()
    opt2.restore()
    opt2.run_resident()
    opt2.sync()
    opt2.download()
    assert np.array_equal(opt2.x, opt.x) and np.array_equal(opt2.traj_costs, opt.traj_costs)


def test_deepcopy_and_getstate():
    import copy
    opt = _factory("lateral_profile")()
    cases.lateral(opt, seed=0)
    opt.update()
    twin = copy.deepcopy(opt)
    assert np.array_equal(twin.x, opt.x) and twin.horizon == opt.horizon
    st = opt.__getstate__()
    assert set(st) == {"u_min", "u_max", "horizon", "opt_start", "barrier_weight", "lg_mult_limit",
                       "max_iterations", "max_lg_iterations", "step", "use_quadratic_terms", "params"}
    twin.update()
    opt.update()
    assert np.array_equal(twin.x, opt.x)


def test_array_shape_assertions_like_the_template():
    """check(ARR.ndim == 1) inside the builtins -> AssertionError after update() (optim.c:375-392, 1195-1200)."""
    opt = _factory("lateral_profile")()
    cases.lateral(opt, seed=0)
    opt.params.k_ref = np.zeros((4, 4))                       # lerp over a 2-D array
    with pytest.raises(AssertionError, match="k_ref.ndim == 1"):
        opt.update()
    fresh = _factory("lateral_profile")()                      # array parameters never set: ndim 0
    with pytest.raises(AssertionError, match="ndim == 1"):
        fresh.update()
    probe = _factory("builtin_probe")()
    cases.builtin_probe(probe, seed=0)
    probe.params.loop_vals = np.zeros(5)                       # lerp_wrap: xs and values differ in length
    with pytest.raises(AssertionError, match="dims"):
        probe.update()


def test_long_horizon_and_per_problem_parameters():
    """T = 299 (the template's capacity) runs with the trajectory mirrors outside LDS;
    per-problem array parameters give every problem of a batch its own reference course."""
    make = _factory("lateral_profile")
    opt = make()
    cases.lateral(opt, seed=0, horizon=299)
    ref = OI.OracleOptim("lateral_profile")
    cases.lateral(ref, seed=0, horizon=299)
    opt.update()
    ref.update()
    assert cases.scaled_err(opt.x, ref.x) < 1e-6 and opt.iterations == ref.iterations
    B = 3
    batch = make(batch=B, horizon=60)
    cases.lateral(batch, seed=0, horizon=60)
    s = np.arange(60) * 0.5
    lowers = np.stack([-2.0 + (1.5 + 0.5 * b) * np.exp(-((s - 12.0 - 3 * b) / 4.0) ** 2) for b in range(B)])
    batch.set_param_per_problem("d_lower_constr", lowers)
    batch.x[...] = 0.0                      # (the set-up function addresses an unbatched object)
    batch.x[:, 0, 0] = 0.1
    batch.u[...] = 0.0
    batch.update()
    for b in range(B):
        one = OI.OracleOptim("lateral_profile")
        cases.lateral(one, seed=0, horizon=60)
        one.params.d_lower_constr = lowers[b]
        one.x[...] = 0.0
        one.x[0, 0] = 0.1
        one.update()
        assert cases.scaled_err(batch.x[b], one.x) < 1e-6, b
        assert int(batch.iterations[b]) == one.iterations


@pytest.mark.parametrize("horizon,t_cap,lds_kb,want_lds", [(37, 37, None, 1), (21, None, None, 1), (21, None, 39, 0),
                                                           (60, 60, None, 1), (13, 20, 8, 0)])
def test_ragged_horizons_match_oracle(horizon, t_cap, lds_kb, want_lds, monkeypatch):
    """Horizons that are not a multiple of the rollout ring (8 stages) and BOTH homes of the trajectory mirrors.
    A single problem has a compute unit to itself, so its mirrors sit in LDS even at the default capacity of 299
    stages (70 KB); TPLHIP_DEBUG=ilqr_lds_kb=N forces the budget of a chip-filling batch (39 KB) or less, which leaves them
    in global memory.  The residency is asserted, not assumed (tplhip_ilqr_info).  60 stages no longer fit the cost
    block into the LDS ring."""
    if lds_kb is not None:
        monkeypatch.setenv("TPLHIP_DEBUG", f"ilqr_lds_kb={lds_kb}")
    make = _factory("trajectory_tracking_mpc")
    opt = make(horizon=horizon, t_cap=t_cap) if t_cap else make()
    ref = OI.OracleOptim("trajectory_tracking_mpc")
    for o in (opt, ref):
        cases.mpc(o, seed=2, curvature=0.01, min_rel=0.0)
        o.horizon = horizon
        o.max_iterations = 6
        o.x[0] = np.array([0.0, 0.3, -0.05, 0.0, 7.0, 0.0, 0.0])
    for _ in range(2):
        opt.update()
        ref.update()
        assert int(opt.iterations) == int(ref.iterations) and float(opt.alpha) == float(ref.alpha)
        assert cases.scaled_err(opt.x[:horizon + 1], ref.x[:horizon + 1]) < 1e-6
        assert cases.scaled_err(opt.u[:horizon], ref.u[:horizon]) < 1e-6
        assert abs(float(opt.traj_costs) - float(ref.traj_costs)) < 1e-8 * abs(float(ref.traj_costs))
    info = opt.info()
    assert info["lds_traj"] == want_lds and info["t_cap"] == (t_cap or 299), info


def test_frozen_prefix_stays_frozen():
    """opt_start > 0.  No tpl caller moves it off 0, and the reference's own line search then starts
    from next_x[optStart], a row nothing ever initialises (optim.c:737-742), and copies that stale
    prefix over x on acceptance -- not a behaviour to reproduce.  Here the stages before opt_start
    are kept, the rollouts start from x[opt_start], and the cost (summed from opt_start like the
    reference's) does not get worse."""
    make = _factory("trajectory_tracking_mpc")
    opt = make(horizon=30)
    cases.mpc(opt, seed=2, curvature=0.01, min_rel=0.0)
    opt.horizon = 30
    opt.max_iterations = 5
    opt.x[0] = np.array([0.0, 0.3, -0.05, 0.0, 7.0, 0.0, 0.0])
    opt.update()                                      # a consistent trajectory to freeze a prefix of
    x_before, u_before, j_before = np.array(opt.x), np.array(opt.u), float(opt.traj_costs)
    opt.opt_start = 4
    opt.update()
    assert np.array_equal(np.asarray(opt.x[:5]), x_before[:5]) and np.array_equal(np.asarray(opt.u[:4]), u_before[:4])
    assert np.isfinite(float(opt.traj_costs)) and float(opt.traj_costs) <= j_before * (1 + 1e-12)
    assert not np.array_equal(np.asarray(opt.x[6:]), x_before[6:])


@pytest.mark.parametrize("name", ["lateral", "velocity_space"])
def test_both_backward_pass_forms_pass_the_goldens_and_agree_bit_for_bit(name, tmp_path):
    """Two-state problems ship with the register-resident backward pass (riccati_backward_small); the lane-parallel
    form every other problem uses is built here for the same problem (-DTPL_ILQR_SMALL_NX=0, hipcc on the box):
    both pass the reference's golden sequence, and their results are the same bits."""
    import shutil
    import subprocess
    from tpl_amd.optim import optimizers as o, solver
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not shutil.which(hipcc) and not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    model, _, kw, _ = cases.CASES[name]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "tpl_amd", "models", f"{model}.inc")
    if not os.path.exists(inc):
        pytest.skip("generated model header not in this tree")
    so = str(tmp_path / f"libilqr_{model}_lane_parallel.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
                           "-fno-fast-math", "-ffp-contract=off", "-DTPL_ILQR_SMALL_NX=0", f'-DTPL_MODEL_INC="{inc}"',
                           "-I", os.path.join(root, "include"), "-o", so,
                           os.path.join(root, "tpl_amd", "csrc", "ilqr", "ilqr_model.hip")], timeout=600)
    cfg = getattr(o, "config_" + model)()
    g = np.load(os.path.join(GOLDEN, f"ilqr_{name}.npz"))
    results = []
    for make in (_factory(model), solver.factory(so, cfg.param_defaults)):
        opt = make()
        cases.apply_inputs(opt, g)
        snaps = cases.run_sequence(opt, int(g["in.shift"]))
        assert not cases.compare(snaps, g), "golden sequence"
        results.append([np.array(getattr(opt, f)) for f in ("x", "u", "k", "K", "lagrange_multiplier")])
    for a, b in zip(*results):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("model,golden", [("trajectory_tracking_mpc_time", "mpc_time"), ("trajectory_tracking_mpc", "mpc_curve")])
def test_other_scheduler_gives_the_same_bits(model, golden, tmp_path, monkeypatch):
    """A legal compiler option must not change a result.  The shipped library was built under one instruction scheduler
    strategy (`info()["scheduler"]`: max-ilp where its listing passes the EXEC = 0 scan, else hipcc's default --
    genopt.compile_model); the same source built under the OTHER strategy on the box must give every field of a ragged batch
    bit for bit, in the single launch and in the fused form -- and mu = 10^(mu_step - 1) on the problems that end on rejected
    steps, the symptom of round 4's max-ilp miscompile (profiles/r05_ilqr_max_ilp_root_cause.txt: AGPR copies of mu executed
    with EXEC = 0; sources that come out of the compiler that way are refused by the scan, tests/test_isa_check.py)."""
    import shutil
    import fuzz_ilqr_forms as F
    from tpl_amd.optim import genopt, solver
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not shutil.which(hipcc) and not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    shipped_runs = {form: F.run(model, golden, 7, form)[0] for form in (0, 2)}
    shipped = _factory(model)().info()["scheduler"]
    other = "default" if shipped == "max-ilp" else "max-ilp"
    so = str(tmp_path / f"libilqr_{other}.so")
    try:
        genopt.compile_model(model, force=True, so_path=so, scheduler=other)
    except RuntimeError as e:
        if "EXEC = 0" in str(e):
            pytest.skip(f"the {other} build of this source does not pass the scan (the compiler bug is live): nothing to compare")
        raise
    monkeypatch.setattr(genopt, "build", lambda cfg, name=None, out_dir=None: solver.factory(so, cfg.param_defaults))
    for form in (0, 2):
        got = F.run(model, golden, 7, form)[0]
        for step, (a, b) in enumerate(zip(shipped_runs[form], got)):
            for f in a:
                assert np.array_equal(a[f], b[f], equal_nan=True), (model, shipped, other, form, step, f)
        mu_step = got[0]["mu_step"]
        assert np.array_equal(got[0]["mu"], np.where(mu_step == 0, 0.0, 10.0 ** (mu_step - 1))), (form, got[0]["mu"])


# ---- split-launch form (k_split_begin / _refresh / _chain / _end): the update() of a chip-filling batch as a sequence of
# launches, so that the dependent chains run with several resident wavefronts per SIMD -- and the fused form
# (k_ilqr_fused): the same phases as calls inside one launch, compiled to the same two-wavefront register budget.  Forced
# here with TPLHIP_DEBUG=ilqr_split=1 / =2 on batches of any size; the same routines in the same order: the same bits.
_FORM_NAMES = {0: "single", 1: "split", 2: "fused", 3: "fused1"}


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("name", list(cases.CASES))
def test_split_launch_passes_the_goldens(name, form, monkeypatch):
    monkeypatch.setenv("TPLHIP_DEBUG", f"ilqr_split={form}")
    model, _, kw, _ = cases.CASES[name]
    g = np.load(os.path.join(GOLDEN, f"ilqr_{name}.npz"))
    opt = _factory(model)()
    cases.apply_inputs(opt, g)
    snaps = cases.run_sequence(opt, int(g["in.shift"]))
    assert opt.info()["form"] == _FORM_NAMES[form], opt.info()
    bad = cases.compare(snaps, g, first_order=bool(kw.get("first_order")))
    assert not bad, "\n".join(bad)


_BIT_FIELDS = ("x", "u", "k", "K", "lagrange_multiplier", "next_x", "next_u", "prev_x", "prev_k", "fx", "fu", "lx", "lu",
               "lxx", "luu", "lux", "traj_costs", "iterations", "lg_iterations", "alpha", "mu", "mu_step",
               "trajectory_changed", "improved", "termination_condition")


@pytest.mark.parametrize("lds_kb", [None, 8, 40])
def test_split_launch_matches_single_launch(lds_kb, monkeypatch):
    """A ragged batch: per-problem iteration budgets and horizons, a convergence threshold that stops problems at
    different iterations, two multiplier rounds.  Both forms, every field and every scalar: the same bits.  Run with the
    split kernels' default LDS budget, a tight one (the mirrors and one of the arrays stay in global memory) and a
    roomy one (arrays and mirrors resident)."""
    make = _factory("trajectory_tracking_mpc")
    B = 70
    rng = np.random.default_rng(5)
    x0 = np.zeros((B, 7))
    x0[:, 1] = rng.uniform(-1, 1, B)
    x0[:, 2] = rng.uniform(-0.2, 0.2, B)
    x0[:, 4] = rng.uniform(4, 10, B)
    results = []
    for form in (0, 1, 2, 3):                           # (3: the fused phases at one wavefront per SIMD, k_ilqr_fused1 -- the A/B leg of round 5)
        dbg = f"ilqr_split={form}" + (f",ilqr_split_lds_kb={lds_kb}" if (form and lds_kb) else "")
        monkeypatch.setenv("TPLHIP_DEBUG", dbg)
        opt = make(batch=B, horizon=50)
        cases.mpc(opt, seed=0, curvature=0.01, min_rel=1e-3)
        opt.x[:, 0] = x0
        opt.max_lg_iterations = 2
        opt.lg_mult_limit = 50.0
        opt.max_iterations = 3 + (np.arange(B) % 6)
        opt.horizon = 50 - (np.arange(B) % 9)
        out = []
        for _ in range(2):
            opt.update()
            out.append({f: np.array(getattr(opt, f)) for f in _BIT_FIELDS})
            opt.shift(1)
        info = opt.info()
        assert info["form"] == _FORM_NAMES[form], info
        if form and lds_kb:
            assert info["lds_traj"] == (1 if lds_kb == 40 else 0) and (info["lds_arrays"] == 5) == (lds_kb == 40), info
        results.append(out)
    for other in results[1:]:
        for a, b in zip(results[0], other):
            for f in _BIT_FIELDS:
                assert np.array_equal(a[f], b[f], equal_nan=True), f
    it = results[0][0]["iterations"]
    assert len(np.unique(it)) > 2, it                  # the batch really is ragged


@pytest.mark.parametrize("model,first_order", [("lateral_profile", False), ("simple_vehicle", True)])
def test_split_launch_matches_single_launch_other_models(model, first_order, monkeypatch):
    """The register-resident backward pass (two-state problems) and the first-order method through the split form."""
    make = _factory(model)
    name = {"lateral_profile": "lateral", "simple_vehicle": "simple_vehicle_first_order"}[model]
    g = np.load(os.path.join(GOLDEN, f"ilqr_{name}.npz"))
    B = 9
    results = []
    for form in (0, 1, 2, 3):
        monkeypatch.setenv("TPLHIP_DEBUG", f"ilqr_split={form}")
        one = make()
        cases.apply_inputs(one, g)
        T = int(one.horizon)
        opt = make(batch=B, horizon=T)
        cases.apply_inputs(opt, g)
        x0 = np.array(one.x[0])
        for b in range(B):
            opt.x[b, 0] = x0 * (1.0 + 0.05 * b)
        opt.update()
        opt.update()
        assert opt.info()["form"] == _FORM_NAMES[form]
        results.append({f: np.array(getattr(opt, f)) for f in _BIT_FIELDS})
    for other in results[1:]:
        for f in _BIT_FIELDS:
            assert np.array_equal(results[0][f], other[f], equal_nan=True), f


def test_forms_fuzz_sample(monkeypatch):
    """A short sample of tests/fuzz_ilqr_forms.py (the long run is a tool): every problem of the goldens as a ragged batch
    with a random integrator and order -- the template parameters of the chain phase -- through all three forms."""
    import fuzz_ilqr_forms
    monkeypatch.setattr(sys, "argv", ["fuzz_ilqr_forms.py", "1"])
    monkeypatch.setenv("TPLHIP_DEBUG", "")
    assert fuzz_ilqr_forms.main() == 0


def test_large_batch_takes_the_fused_form_and_matches_oracle():
    """4 096 problems (four per SIMD): update() picks the two-wavefront fused form by itself; every 256th problem is
    re-solved by the CPU oracle."""
    make = _factory("trajectory_tracking_mpc")
    B = 4096
    opt = make(batch=B, horizon=50)
    cases.mpc(opt, seed=0, min_rel=0.0)
    rng = np.random.default_rng(1)
    x0 = np.zeros((B, 7))
    x0[:, 1] = rng.uniform(-1, 1, B)
    x0[:, 2] = rng.uniform(-0.2, 0.2, B)
    x0[:, 4] = rng.uniform(4, 10, B)
    opt.x[:, 0] = x0
    opt.max_iterations = 8
    opt.update()
    assert opt.info()["form"] == "fused", opt.info()
    assert np.all(opt.iterations == 8) and np.all(np.isfinite(opt.traj_costs))
    for b in range(0, B, 256):
        ref = OI.OracleOptim("trajectory_tracking_mpc")
        cases.mpc(ref, seed=0, min_rel=0.0)
        ref.max_iterations = 8
        ref.x[0] = x0[b]
        ref.update()
        assert cases.scaled_err(opt.x[b], ref.x) < 1e-6, b
        assert cases.scaled_err(opt.u[b], ref.u) < 1e-6, b
        assert abs(opt.traj_costs[b] - ref.traj_costs) < 1e-8 * abs(ref.traj_costs), b


def test_small_models_take_the_fused_form_for_large_batches_too(monkeypatch):
    """Not only the tracking MPCs: every model whose single launch is held to one wavefront per SIMD gains from the fused
    form once the batch exceeds what that launch keeps resident (tools/ab_ilqr_forms_all.py, profiles/r04_ilqr_split_ab.txt:
    lateral_profile +22 ... +34 %).  A large batch of lateral_profile takes it by itself, and every 128th problem matches the
    single launch bit for bit."""
    make = _factory("lateral_profile")
    g = np.load(os.path.join(GOLDEN, "ilqr_lateral.npz"))
    outs = []
    for dbg in ("", "ilqr_split=0"):
        monkeypatch.setenv("TPLHIP_DEBUG", dbg)
        opt = make(batch=2048, horizon=int(g["in.horizon"]))
        cases.apply_inputs(opt, g)
        opt.x[:, 0, 0] += 0.001 * np.arange(2048)
        opt.update()
        outs.append((opt.info()["form"], np.array(opt.x[::128]), np.array(opt.traj_costs)))
    assert outs[0][0] == "fused" and outs[1][0] == "single", (outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert np.all(np.isfinite(outs[0][2]))
