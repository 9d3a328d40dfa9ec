#!/usr/bin/env python3
"""Generates tests/golden/pvd_*.npz by running the REFERENCE's path/velocity-decomposition glue (imported from
/root/reference, which exists in the build container only) on the synthetic planning cycles of tests/pvd_cases.py.

    CFLAGS="-DPyArray_MoveInto=PyArray_CopyInto" python tests/golden/make_pvd_goldens.py

What runs is the reference's own Python: tpl.planning.utils.rampify_profile, tpl.planning.path_vel_decomp
{path_optim.rampify_profile, PathOptim, PathSmoothing, VelocityOptim}, tpl.util.{resample_path, short_angle_dist,
normalize_angle}, tpl.environment.map_module.{add_vel_constraint, zero_after_first_zero}, driving the reference's
genopt CPU solvers (built here like tests/golden/make_ilqr_goldens.py does).  Those modules import packages this
image lacks; the SHIMS below only satisfy their import statements and decorators:
  * numba        -> ``njit`` is the identity decorator: the ramp functions run as the plain Python they are
                    (numba's fastmath may reassociate; the reference is only defined up to that: "weakly pinned");
  * objtoolbox   -> ``merge(dst, src)`` copies src's attributes onto dst (its documented effect on flat objects);
                    ``bundle`` an empty namespace;
  * structstore  -> never called here;
  * tplcpp       -> ``project`` / ``resample`` / ``Projection`` are THIS repository's restatement
                    (tpl_amd.planning.geometry, from library/src/utils.cpp:257-548; the C++ extension needs CUDA +
                    Eigen + pybind11 and cannot be built here), every other name an inert stand-in.
So the fixtures pin the glue LOGIC and the ramp functions against the reference's Python; the two geometry
routines are the same code on both sides (their own checks: tests/test_pvd_glue.py, hand-derived cases).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ.setdefault("CFLAGS", "-DPyArray_MoveInto=PyArray_CopyInto")


class _Inert:
    """Stand-in for names that are imported but never used on this path."""

    def __getattr__(self, k):
        return _Inert()

    def __call__(self, *a, **k):
        return _Inert()

    def __sub__(self, o):
        return 0

    def __index__(self):
        return 0

    def __gt__(self, o):
        return False

    def __lt__(self, o):
        return False


def install_shims():
    from tpl_amd.planning import geometry as geo
    numba = types.ModuleType("numba")
    numba.njit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
    sys.modules["numba"] = numba
    otb = types.ModuleType("objtoolbox")
    otb.merge = lambda dst, src: [setattr(dst, k, getattr(src, k)) for k in vars(src) if not k.startswith("_")] and None
    otb.bundle = lambda: types.SimpleNamespace()
    otb.get_obj_dict = lambda o: {k: v for k, v in vars(o).items() if not k.startswith("_")}
    sys.modules["objtoolbox"] = otb
    sys.modules["structstore"] = types.ModuleType("structstore")
    tplcpp = types.ModuleType("tplcpp")
    tplcpp.__getattr__ = lambda name: _Inert()
    tplcpp.project = geo.project
    tplcpp.resample = geo.resample
    tplcpp.Projection = geo.Projection
    sys.modules["tplcpp"] = tplcpp


def main():
    install_shims()
    sys.path.insert(0, "/root/reference/library")
    import pvd_cases as cases
    from tpl.optim import genopt, optimizers as ref_opts
    import tpl.util as rutil
    import tpl.environment.map_module as rmm
    from tpl.planning.utils import rampify_profile as ref_ramp_v
    import tpl.planning.path_vel_decomp.path_optim as rpo
    import tpl.planning.path_vel_decomp.path_smoothing as rps
    import tpl.planning.path_vel_decomp.velocity_optim as rvo

    for name in ("lateral_profile", "velocity_profile_space", "ref_line_smoother_k"):
        cfg = getattr(ref_opts, "config_" + name)()
        cfg.output_dir = "/tmp/genopt_oracle/"
        setattr(ref_opts, name, genopt.build(cfg))

    out = {}
    # ---- functions
    for i, c in enumerate(cases.rampify_velocity_cases()):
        out[f"ramp_v{i}"] = ref_ramp_v(c["v0"], c["a0"], c["lim_v"].copy(), c["a_min"], c["a_max"], c["j_min"], c["j_max"], c["v_min"], c["step"])
    for i, c in enumerate(cases.rampify_lateral_cases()):
        out[f"ramp_l{i}"] = rpo.rampify_profile(c["step"], c["horizon"], c["evasion_sharpness"], c["proj_distance"], c["path"],
                                                 c["gap"], c["lower"], c["upper"])
    ang = np.linspace(-9.0, 9.0, 37)
    out["normalize_angle"] = np.array([rutil.normalize_angle(a) for a in ang])
    out["short_angle_dist"] = np.array([[rutil.short_angle_dist(a, b) for b in ang[::4]] for a in ang])
    path = cases.ref_path(120)
    out["resample_path"] = rutil.resample_path(path, 0.7, 60, start_index=3, zero_vel_at_end=True)
    out["resample_path_long"] = rutil.resample_path(path[:30], 0.5, 40)          # runs past the end: extrapolation branch
    lim = np.full(50, 9.0)
    rmm.add_vel_constraint(lim, 12, 3.0, length=20)
    rmm.add_vel_constraint(lim, -4, 5.0, length=10)
    rmm.add_vel_constraint(lim, 45, 1.0, 10, 2)
    out["add_vel_constraint"] = lim
    out["zero_after_first_zero"] = rmm.zero_after_first_zero(np.array([1.0, 1, 1, 0, 1, 1, 0, 1]))

    # ---- the three solver users over three consecutive planning cycles (reset, then two warm starts)
    ps, po, vo = rps.PathSmoothing(), rpo.PathOptim(), rvo.VelocityOptim()
    vo_s = rvo.VelocityOptim()
    for cyc in range(3):
        env = cases.environment(cyc)
        ps.update(env, cases.default_params("path_smoothing"))
        out[f"c{cyc}.smooth.opt_path"] = ps.opt_path.copy()
        out[f"c{cyc}.smooth.iterations"] = np.array(ps.opt.iterations)
        po.update(env, cases.default_params("path_optim"))
        out[f"c{cyc}.optim.opt_path"] = po.opt_path.copy()
        out[f"c{cyc}.optim.d_lower_constr"] = po.d_lower_constr.copy()
        out[f"c{cyc}.optim.d_upper_constr"] = po.d_upper_constr.copy()
        out[f"c{cyc}.optim.d_lower_ref"] = np.array(po.opt.params.d_lower_constr)
        out[f"c{cyc}.optim.d_upper_ref"] = np.array(po.opt.params.d_upper_constr)
        out[f"c{cyc}.optim.d_offset"] = np.array(po.opt.params.d_offset)
        out[f"c{cyc}.optim.x"] = np.array(po.opt.x)
        out[f"c{cyc}.optim.iterations"] = np.array(po.opt.iterations)
        vo.update(po.opt_path, env, cases.default_params("velocity_optim"))
        for k in ("v_opt", "v_lim", "v_ref", "stop_mask"):
            out[f"c{cyc}.vel.{k}"] = np.array(getattr(vo, k), dtype=np.float64)
        out[f"c{cyc}.vel.leader"] = np.array([vo.s_leader, vo.v_leader])
        out[f"c{cyc}.vel.x"] = np.array(vo.opt.x)
        out[f"c{cyc}.vel.u"] = np.array(vo.opt.u)
        out[f"c{cyc}.vel.ref_v_weight"] = np.array(vo.opt.params.ref_v_weight)
        out[f"c{cyc}.vel.ref_t_min"] = np.array(vo.opt.params.ref_t_min)
        out[f"c{cyc}.vel.iterations"] = np.array(vo.opt.iterations)
        # the smoothed path through the velocity optimisation too (the planner's smooth_only branch)
        vo_s.update(ps.opt_path, cases.environment(cyc, with_objects=False), cases.default_params("velocity_optim"))
        out[f"c{cyc}.vel_smooth.v_opt"] = np.array(vo_s.v_opt)
    # ---- the planner plugin on top: three cycles with obstacle avoidance, three smoothing only
    import copy as _copy
    from tpl.planning.path_vel_decomp.path_vel_decomp_planner import PathVelDecompPlanner

    class _Shared:
        def __setattr__(self, k, v):
            if k == "params" and not hasattr(v, "deepcopy"):
                v.deepcopy = lambda v=v: _copy.deepcopy(v)
            object.__setattr__(self, k, v)

    from contextlib import contextmanager

    @contextmanager
    def _lock():
        yield
    for tag, smooth_only in (("plan", False), ("plan_smooth", True)):
        sh = _Shared()
        pl = PathVelDecompPlanner(sh, _lock)
        sh.params.smooth_only = smooth_only
        sh.params.write_debug_data = not smooth_only       # the debug block reads path-optimisation results
        for k, v in vars(cases.default_params("path_optim")).items():
            setattr(sh.params.path_optim, k, v)
        for k, v in vars(cases.default_params("velocity_optim")).items():
            setattr(sh.params.velocity_optim, k, v)
        for k, v in vars(cases.default_params("path_smoothing")).items():
            setattr(sh.params.path_smoothing, k, v)
        for cyc in range(3):
            sh_env = cases.shared_environment(cyc)
            tr = pl.update(sh_env)
            out[f"{tag}.c{cyc}.traj"] = np.stack([tr.time, tr.s, tr.x, tr.y, tr.orientation, tr.curvature, tr.velocity,
                                                  np.ravel(tr.acceleration)], axis=1)
            out[f"{tag}.c{cyc}.requests"] = np.array([float(sh_env.local_map.update_inters_paths),
                                                      sh_env.local_map.step_shift_idx, sh_env.local_map.shift_vel_lim])
            out[f"{tag}.c{cyc}.leader"] = np.array([sh.debug.s_leader, sh.debug.v_leader])
            if not smooth_only:
                out[f"{tag}.c{cyc}.dbg_v_ref"] = np.array(sh.debug.v_ref)
                out[f"{tag}.c{cyc}.dbg_path_lower_constr"] = np.array(sh.debug.path_lower_constr)
    dst = os.path.join(HERE, "pvd_glue.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays,", os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
