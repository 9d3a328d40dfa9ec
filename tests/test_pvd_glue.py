"""The path/velocity-decomposition glue (tpl_amd/planning/) against fixtures captured from the
REFERENCE's own Python glue driving the reference's CPU solvers (tests/golden/pvd_glue.npz, made by
tests/golden/make_pvd_goldens.py; its header states the import shims: the ramp functions are "weakly pinned" --
numba's fastmath is the identity there -- and the two geometry routines are this repository's restatement on both
sides, so they get hand-derived checks of their own below).  CPU: helpers, and the three solver users on the CPU
oracle solvers.  GPU: the same three on the HIP solvers."""
import os

import numpy as np
import pytest

import oracle_ilqr as OI
import pvd_cases as cases
from tpl_amd.planning import geometry as geo, utils as putils
from tpl_amd.planning.path_vel_decomp import PathOptim, PathSmoothing, VelocityOptim, rampify_lateral_profile

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pvd_glue.npz"))


# ---- geometry: hand-derived ----------------------------------------------------------------------

def test_project_on_a_polyline_by_hand():
    pts = np.array([[0.0, 0.0], [4.0, 0.0], [4.0, 3.0]])
    p = geo.project(pts, [1.0, 2.0])                         # above the first segment: left of it -> positive... sign rule below
    assert (p.start, p.end, p.index) == (0, 1, 0) and p.alpha == 0.25 and p.in_bounds
    assert np.allclose(p.point, [1.0, 0.0]) and p.arc_len == 1.0 and abs(p.distance) == 2.0
    q = geo.project(pts, [1.0, -2.0])
    assert q.distance == -p.distance                         # the two sides of the strip have opposite signs
    r = geo.project(pts, [5.0, 2.0])                          # beside the second segment, past the middle
    assert (r.start, r.end, r.index) == (1, 2, 2) and np.isclose(r.alpha, 2.0 / 3.0) and np.isclose(r.arc_len, 6.0)
    assert np.isclose(r.angle, np.pi / 2) and np.sign(r.distance) == np.sign(q.distance)      # right of the strip both times
    s = geo.project(pts, [-2.0, 1.0])                         # before the start: clamped, out of bounds
    assert not s.in_bounds and s.alpha == 0.0 and np.allclose(s.point, [0.0, 0.0])
    e = geo.project(pts, [4.0, 9.0])                          # past the end
    assert not e.in_bounds and e.alpha == 1.0 and np.isclose(e.arc_len, 7.0)
    many = geo.project(pts, np.array([[1.0, 2.0], [5.0, 2.0]]))
    assert [m.index for m in many] == [0, 2]
    loop = np.array([[0.0, 0.0], [4.0, 0.0], [4.0, 4.0], [0.0, 4.0]])
    c = geo.project(loop, [-1.0, 2.0], closed=True)           # on the closing segment (3 -> 0)
    assert (c.start, c.end) == (3, 0) and np.isclose(c.alpha, 0.5) and np.isclose(c.arc_len, 14.0)
    with pytest.raises(ValueError):
        geo.project(np.zeros((1, 2)), [0.0, 0.0])


def test_resample_equidistant_by_hand():
    pts = np.array([[0.0, 0.0], [3.0, 0.0], [3.0, 4.0]])
    rs = geo.resample(pts, 1.0, 6)
    assert np.allclose(rs[:4, :2], [[0, 0], [1, 0], [2, 0], [3, 0]]) and np.allclose(rs[:4, 2], [0, 1 / 3, 2 / 3, 1])
    assert np.allclose(np.linalg.norm(np.diff(rs[:, :2], axis=0), axis=1), 1.0)           # chord length = sampling distance
    assert np.allclose(rs[4:, :2], [[3, 1], [3, 2]]) and list(rs[4:, 3]) == [1, 1] and list(rs[4:, 4]) == [2, 2]
    far = geo.resample(pts, 1.0, 12)                          # past the end: extrapolates along the last segment
    assert np.allclose(far[-1, :2], [3.0, 8.0]) and far[-1, 2] > 1.0
    dup = geo.resample(np.array([[0.0, 0.0], [0.0, 0.0], [2.0, 0.0]]), 0.5, 3)           # duplicate points are dropped
    assert np.allclose(dup[:, 0], [0, 0.5, 1.0])
    assert geo.resample(np.array([[1.0, 2.0]]), 1.0, 5).shape == (1, 5)
    assert geo.resample(pts, 1.0, 0).shape == (0, 5)


# ---- functions against the reference's Python ------------------------------------------------------

def test_ramp_functions_match_the_reference():
    for i, c in enumerate(cases.rampify_velocity_cases()):
        got = putils.rampify_profile(c["v0"], c["a0"], c["lim_v"].copy(), c["a_min"], c["a_max"], c["j_min"], c["j_max"],
                                     c["v_min"], c["step"])
        assert np.allclose(got, GOLD[f"ramp_v{i}"], rtol=1e-12, atol=1e-12), i
    for i, c in enumerate(cases.rampify_lateral_cases()):
        got = rampify_lateral_profile(c["step"], c["horizon"], c["evasion_sharpness"], c["proj_distance"], c["path"],
                                         c["gap"], c["lower"], c["upper"])
        assert np.allclose(got, GOLD[f"ramp_l{i}"], rtol=1e-12, atol=1e-12), i
        assert np.all(got[:c["horizon"]] >= c["lower"] - 1e-12) and np.all(got[c["horizon"]:] == -10.0)


def test_angle_path_and_limit_helpers_match_the_reference():
    ang = np.linspace(-9.0, 9.0, 37)
    assert np.allclose([geo.normalize_angle(a) for a in ang], GOLD["normalize_angle"], atol=1e-15)
    assert np.allclose(geo.normalize_angle(ang), GOLD["normalize_angle"], atol=1e-15)
    assert np.allclose(geo.short_angle_dist(ang[:, None], ang[None, ::4]), GOLD["short_angle_dist"], atol=1e-14)
    path = cases.ref_path(120)
    assert np.allclose(geo.resample_path(path, 0.7, 60, start_index=3, zero_vel_at_end=True), GOLD["resample_path"], atol=1e-12)
    assert np.allclose(geo.resample_path(path[:30], 0.5, 40), GOLD["resample_path_long"], atol=1e-12)
    lim = np.full(50, 9.0)
    geo.add_vel_constraint(lim, 12, 3.0, length=20)
    geo.add_vel_constraint(lim, -4, 5.0, length=10)
    geo.add_vel_constraint(lim, 45, 1.0, 10, 2)
    assert np.array_equal(lim, GOLD["add_vel_constraint"])
    assert np.array_equal(geo.zero_after_first_zero(np.array([1.0, 1, 1, 0, 1, 1, 0, 1])), GOLD["zero_after_first_zero"])
    assert np.array_equal(geo.zero_after_first_zero(np.ones(4)), np.ones(4))


# ---- the three solver users ------------------------------------------------------------------------

def _run_cycles(make):
    """Three consecutive planning cycles (reset, two warm starts) like the fixture generator; ``make(problem)``
    returns the solver object each glue class drives."""
    ps, po = PathSmoothing(make("ref_line_smoother_k")), PathOptim(make("lateral_profile"))
    vo, vo_s = VelocityOptim(make("velocity_profile_space")), VelocityOptim(make("velocity_profile_space"))
    res = {}
    for cyc in range(3):
        env = cases.environment(cyc)
        ps.update(env, cases.default_params("path_smoothing"))
        res[f"c{cyc}.smooth.opt_path"] = ps.opt_path.copy()
        res[f"c{cyc}.smooth.iterations"] = np.array(ps.opt.iterations)
        po.update(env, cases.default_params("path_optim"))
        res[f"c{cyc}.optim.opt_path"] = po.opt_path.copy()
        res[f"c{cyc}.optim.d_lower_constr"] = po.d_lower_constr.copy()
        res[f"c{cyc}.optim.d_upper_constr"] = po.d_upper_constr.copy()
        res[f"c{cyc}.optim.d_lower_ref"] = np.array(po.opt.params.d_lower_constr)
        res[f"c{cyc}.optim.d_upper_ref"] = np.array(po.opt.params.d_upper_constr)
        res[f"c{cyc}.optim.d_offset"] = np.array(po.opt.params.d_offset)
        res[f"c{cyc}.optim.x"] = np.array(po.opt.x)
        res[f"c{cyc}.optim.iterations"] = np.array(po.opt.iterations)
        vo.update(po.opt_path, env, cases.default_params("velocity_optim"))
        for k in ("v_opt", "v_lim", "v_ref", "stop_mask"):
            res[f"c{cyc}.vel.{k}"] = np.array(getattr(vo, k), dtype=np.float64)
        res[f"c{cyc}.vel.leader"] = np.array([vo.s_leader, vo.v_leader])
        res[f"c{cyc}.vel.x"] = np.array(vo.opt.x)
        res[f"c{cyc}.vel.u"] = np.array(vo.opt.u)
        res[f"c{cyc}.vel.ref_v_weight"] = np.array(vo.opt.params.ref_v_weight)
        res[f"c{cyc}.vel.ref_t_min"] = np.array(vo.opt.params.ref_t_min)
        res[f"c{cyc}.vel.iterations"] = np.array(vo.opt.iterations)
        vo_s.update(ps.opt_path, cases.environment(cyc, with_objects=False), cases.default_params("velocity_optim"))
        res[f"c{cyc}.vel_smooth.v_opt"] = np.array(vo_s.v_opt)
    assert vo.update.runtime > 0.0 and po.update.runtime > 0.0                # the @runtime surface (util.py:54-67)
    return res


def _compare(res, tol):
    worst = {}
    for k, got in res.items():
        want = GOLD[k]
        assert np.shape(np.squeeze(got)) == np.shape(np.squeeze(want)), (k, np.shape(got), np.shape(want))
        if k.endswith("iterations"):
            assert np.array_equal(np.squeeze(got), np.squeeze(want)), (k, got, want)
            continue
        scale = max(1.0, float(np.max(np.abs(want))))
        err = float(np.max(np.abs(np.squeeze(got) - np.squeeze(want)))) / scale
        worst[k] = err
        assert err <= tol, (k, err)
    return worst


def test_glue_on_the_cpu_oracle_solvers_matches_the_reference_glue():
    res = _run_cycles(lambda problem: OI.OracleOptim(problem))
    worst = _compare(res, 1e-6)
    # what the scenario exercises: an evasion (the corridor really moves), a leader and a time constraint
    assert np.max(res["c0.optim.d_lower_constr"]) > np.max(-cases.environment(0).local_map.d_right + 0.95) + 0.5
    assert res["c0.vel.leader"][1] > 4.0 and res["c0.vel.leader"][0] < 100.0
    assert np.count_nonzero(res["c0.vel.ref_t_min"]) == 1 and np.min(res["c0.vel.ref_v_weight"]) < 1.0
    assert max(worst.values()) < 1e-6


def _run_planner(make):
    """The planner plugin over the same three cycles, with and without obstacle avoidance (fixture: the reference's
    PathVelDecompPlanner on the reference's solvers)."""
    from contextlib import contextmanager
    from tpl_amd.planning.path_vel_decomp import PathVelDecompPlanner
    from tpl_amd.store import Bundle

    @contextmanager
    def lock():
        yield
    res = {}
    for tag, smooth_only in (("plan", False), ("plan_smooth", True)):
        sh = Bundle()
        pl = PathVelDecompPlanner(sh, lock, solvers={k: make(k) for k in ("ref_line_smoother_k", "lateral_profile",
                                                                          "velocity_profile_space")})
        sh.params.smooth_only = smooth_only
        sh.params.write_debug_data = not smooth_only
        for kind in ("path_optim", "velocity_optim", "path_smoothing"):
            for k, v in vars(cases.default_params(kind)).items():
                setattr(getattr(sh.params, kind), k, v)
        for cyc in range(3):
            sh_env = cases.shared_environment(cyc)
            tr = pl.update(sh_env)
            res[f"{tag}.c{cyc}.traj"] = np.stack([tr.time, tr.s, tr.x, tr.y, tr.orientation, tr.curvature, tr.velocity,
                                                  np.ravel(tr.acceleration)], axis=1)
            res[f"{tag}.c{cyc}.requests"] = np.array([float(sh_env.local_map.update_inters_paths),
                                                      sh_env.local_map.step_shift_idx, sh_env.local_map.shift_vel_lim])
            res[f"{tag}.c{cyc}.leader"] = np.array([sh.debug.s_leader, sh.debug.v_leader])
            if not smooth_only:
                res[f"{tag}.c{cyc}.dbg_v_ref"] = np.array(sh.debug.v_ref)
                res[f"{tag}.c{cyc}.dbg_path_lower_constr"] = np.array(sh.debug.path_lower_constr)
        assert pl.update.runtime > 0.0
    return res


def test_planner_plugin_on_the_cpu_oracle_solvers_matches_the_reference_planner():
    res = _run_planner(lambda problem: OI.OracleOptim(problem))
    _compare(res, 1e-6)
    assert res["plan.c0.traj"].shape == (250, 8) and res["plan.c0.leader"][1] > 4.0


@pytest.mark.gpu
def test_planner_plugin_on_the_hip_solvers_matches_the_reference_planner():
    from tpl_amd.optim import optimizers as opts
    opts.build_optimizers()
    _compare(_run_planner(lambda problem: getattr(opts, problem)()), 1e-5)


@pytest.mark.gpu
def test_glue_on_the_hip_solvers_matches_the_reference_glue():
    from tpl_amd.optim import optimizers as opts
    opts.build_optimizers()
    res = _run_cycles(lambda problem: getattr(opts, problem)())
    _compare(res, 1e-5)
    # and the default construction path (no solver handed in) builds its own HIP solver
    po = PathOptim()
    po.update(cases.environment(0), cases.default_params("path_optim"))
    assert np.allclose(po.opt_path, GOLD["c0.optim.opt_path"], atol=1e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference/library/tpl"), reason="the reference checkout exists in the build container only")
def test_the_references_own_planner_file_runs_on_these_stage_classes():
    """INTEGRATION.md's claim, executed: tpl's path_vel_decomp_planner.py UNCHANGED, with PathSmoothing / PathOptim /
    VelocityOptim swapped for tpl_amd's (on the CPU restatement of the solvers), reproduces the fixture its own
    classes produced.  Runs in a subprocess with the import shims of the fixture generator."""
    import subprocess
    import sys
    code = r'''
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import make_pvd_goldens as shims
shims.install_shims()
sys.modules["objtoolbox"].get_obj_dict = lambda o: {k: v for k, v in vars(o).items() if not k.startswith("_")}
sys.path.insert(0, "/root/reference/library")
import pvd_cases as cases
import oracle_ilqr as OI
import tpl.planning.path_vel_decomp.path_vel_decomp_planner as rp
from tpl_amd.planning.path_vel_decomp import PathOptim, PathSmoothing, VelocityOptim
rp.PathSmoothing = lambda: PathSmoothing(OI.OracleOptim("ref_line_smoother_k"))
rp.PathOptim = lambda: PathOptim(OI.OracleOptim("lateral_profile"))
rp.VelocityOptim = lambda: VelocityOptim(OI.OracleOptim("velocity_profile_space"))
from contextlib import contextmanager
@contextmanager
def lock():
    yield
class Shared:
    def __setattr__(self, k, v):
        if k == "params" and not hasattr(v, "deepcopy"):
            v.deepcopy = lambda v=v: copy.deepcopy(v)
        object.__setattr__(self, k, v)
gold = np.load(os.path.join(ROOT, "tests", "golden", "pvd_glue.npz"))
sh = Shared()
pl = rp.PathVelDecompPlanner(sh, lock)
sh.params.write_debug_data = True
for kind in ("path_optim", "velocity_optim", "path_smoothing"):
    for k, v in vars(cases.default_params(kind)).items():
        setattr(getattr(sh.params, kind), k, v)
for cyc in range(3):
    tr = pl.update(cases.shared_environment(cyc))
    rows = np.stack([tr.time, tr.s, tr.x, tr.y, tr.orientation, tr.curvature, tr.velocity, np.ravel(tr.acceleration)], axis=1)
    want = gold[f"plan.c{cyc}.traj"]
    err = np.max(np.abs(rows - want)) / max(1.0, np.max(np.abs(want)))
    assert err < 1e-6, (cyc, err)
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + code], cwd=root, capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, PYTHONPATH=root, CFLAGS="-DPyArray_MoveInto=PyArray_CopyInto"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
