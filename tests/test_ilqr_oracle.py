"""CPU tests: the HP-2 oracle (oracle/ilqr_oracle.cpp + the build's own generated
model code) against the golden vectors captured from the reference genopt
solver (tests/golden/ilqr_*.npz, made by tests/golden/make_ilqr_goldens.py)."""
import os

import numpy as np
import pytest

import ilqr_cases as cases
import oracle_ilqr as OI

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden(name):
    return np.load(os.path.join(GOLDEN, f"ilqr_{name}.npz"))


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference_golden(name):
    model, _, kw, _ = cases.CASES[name]
    g = _golden(name)
    opt = OI.OracleOptim(model)
    cases.apply_inputs(opt, g)
    # dynamics()/ct_dynamics() spot values (optim.c:1496-1652)
    assert np.allclose(opt.dynamics(g["dyn.x"], g["dyn.u"], 3, 0.07), g["dyn.dynamics"], rtol=1e-12, atol=1e-14)
    assert np.allclose(opt.ct_dynamics(g["dyn.x"], g["dyn.u"], 3, 0.07), g["dyn.ct_dynamics"], rtol=1e-12, atol=1e-14)
    snaps = cases.run_sequence(opt, int(g["in.shift"]))
    bad = cases.compare(snaps, g, first_order=bool(kw.get("first_order")))
    assert not bad, "\n".join(bad)


def test_setup_functions_reproduce_golden_inputs():
    """The fixtures' "in.*" part is what tests/ilqr_cases.py sets up (so the HIP
    tests, which configure through the same functions, start from the same inputs)."""
    for name, (model, setup, kw, shift) in cases.CASES.items():
        g = _golden(name)
        opt = OI.OracleOptim(model)
        setup(opt, **kw)
        for f in cases.INPUT_ARRAYS:
            assert np.array_equal(np.asarray(getattr(opt, f)), g["in." + f]), (name, f)
        for p in opt.params.__slots__:
            assert np.array_equal(np.asarray(getattr(opt.params, p)), g["in.param." + p]), (name, p)


def test_lerp_border_behaviour_is_the_references():
    """The reference clamps (size_t)floor(q): below the table start the lookup
    returns the LAST sample (optim.c:347-355).  The MPC derivative code hits this
    at s_r = 0 (finite difference at -5e-5); pinned by the goldens above, stated
    explicitly here through ct_dynamics of the lateral model (lerp over t*dt)."""
    opt = OI.OracleOptim("trajectory_tracking_mpc")
    cases.mpc(opt, seed=0, curvature=0.02)
    x = np.array(opt.x[0], dtype=np.float64)
    x[5] = -0.25                       # half a ref_step before the table start
    f_neg = opt.ct_dynamics(x, np.zeros(2), 0, 0.05)
    x[5] = 99 * 0.5                    # last sample
    f_last = opt.ct_dynamics(x, np.zeros(2), 0, 0.05)
    # s_dot uses r_phi, r_x, r_y, r_k looked up at s_r: identical at both ends
    assert f_neg[5] == pytest.approx(f_last[5], rel=1e-12)


def test_default_values_match_pyOptimAlloc():
    # optim.c:1894-1921
    opt = OI.OracleOptim("lateral_profile")
    assert (opt.dt, opt.horizon, opt.max_iterations, opt.max_lg_iterations) == (0.05, 20, 5, 1)
    assert opt.min_rel_cost_change == 1e-6 and opt.use_quadratic_terms is True
    assert np.all(np.isinf(opt.lg_mult_limit)) and np.all(opt.barrier_weight == 1.0)
    assert np.all(opt.u_max == np.inf) and np.all(opt.u_min == -np.inf)
    opt.horizon = 30                   # rows past the first 20 were never initialised: zeros
    assert np.all(opt.u_max[20:] == 0.0)
    opt.horizon = 1000                 # clamped to H_MAX-1 (optim.c:1726-1734)
    assert opt.horizon == 299


def test_quad_cooperative_dynamics_are_emitted_where_they_pay():
    """The generator emits ctDynamicsQ (look-ups and sin/cos spread over a quad of lanes) for models
    whose dynamics hold at least two independent look-ups or trigonometric evaluations, inside the
    TPL_QUAD_ROUTINES guard (the CPU oracle never sees it), and nothing for the others."""
    from tpl_amd.optim import genopt, optimizers as o
    txt = genopt.emit_model(o.config_trajectory_tracking_mpc(), "trajectory_tracking_mpc")
    guard = txt.index("#if defined(TPL_QUAD_ROUTINES)")
    q = txt[guard:]
    assert "#define TPL_HAS_QUAD_DYNAMICS 1" in q and "void ctDynamicsQ(" in q
    assert q.count("tpl_quad_lookup(") == 1            # ref_phi, ref_k, ref_x, ref_y at x[5]: one batch
    assert q.count("tpl_sincos(qarg") == 1             # sin/cos of x[2], of the reference heading, tan(x[3])
    assert "tpl_tan_sc(x[3]" in q and "tpl_lerp(" not in q
    assert "ctDynamicsQ" not in txt[:guard]
    plain = genopt.emit_model(o.config_velocity_profile_time(), "velocity_profile_time")
    assert "TPL_HAS_QUAD_DYNAMICS" not in plain and "ctDynamicsQ" not in plain


def test_lookups_at_one_abscissa_are_batched_in_every_routine():
    """Look-ups that share (x0, dx, x) come out as ONE tpl_lerp_batch (one division, one memory
    round trip); a lone look-up stays the plain builtin.  The oracle's tpl_lerp_batch is nothing but
    its members, so the golden cases check the batched product against the unbatched arithmetic."""
    import re
    from tpl_amd.optim import genopt, optimizers as o
    txt = genopt.emit_model(o.config_trajectory_tracking_mpc(), "trajectory_tracking_mpc")
    plain = txt[:txt.index("#if defined(TPL_QUAD_ROUTINES)")]
    costs = plain[plain.index("void costs("):plain.index("void stateGradient(")]
    assert costs.count("tpl_lerp_batch<5>(0.0, P.ref_step, x[5]") == 1      # ref_phi/x/y/k/v: one batch
    assert "tpl_lerp(" not in costs and "tpl_lerp_angle(" not in costs
    for m in re.finditer(r"const DynArr\* const qa\[(\d+)\] = \{([^}]*)\};\s*double qo\[(\d+)\];\s*tpl_lerp_batch<(\d+)>", plain):
        n = int(m.group(1))
        assert n >= 2 and n == int(m.group(3)) == int(m.group(4)) and m.group(2).count("&P.") == n


def test_generated_text_does_not_depend_on_the_hash_seed():
    """Batches are collected from sympy sets; their member order must not follow the interpreter's
    hash seed (the committed models/<problem>.inc has to be reproducible)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import hashlib, sys; sys.path.insert(0, %r); "
            "from tpl_amd.optim import genopt, optimizers as o; "
            "print(hashlib.sha1(genopt.emit_model(o.config_trajectory_tracking_mpc(), 'trajectory_tracking_mpc').encode()).hexdigest())" % root)
    digests = set()
    for seed in ("1", "2", "77"):
        env = dict(os.environ, PYTHONHASHSEED=seed)
        digests.add(subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True, text=True).stdout.strip())
    assert len(digests) == 1, digests
    built = os.path.join(root, "tpl_amd", "models", "trajectory_tracking_mpc.inc")
    if os.path.exists(built):                                                    # (generated by build())
        import hashlib
        with open(built) as fd:
            assert hashlib.sha1(fd.read().encode()).hexdigest() in digests       # the text that is built
