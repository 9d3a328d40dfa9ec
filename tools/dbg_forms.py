"""Which fields differ between the launch forms of the batched iLQR update() on one ragged batch of tests/fuzz_ilqr_forms.py
(the case that exposed the max-ilp miscompile of round 4): python tools/dbg_forms.py [path of a solver library build]"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import fuzz_ilqr_forms as F
outs = {}
from tpl_amd.optim import genopt, optimizers as o, solver
if len(sys.argv) > 1:
    lib = os.path.abspath(sys.argv[1])
    genopt.build = lambda cfg, name=None, out_dir=None: solver.factory(lib, cfg.param_defaults)
for form in (0, 1, 2):
    out, what = F.run("trajectory_tracking_mpc_time", "mpc_time", 7, form)
    outs[form] = out
    print(form, what, "mu", out[0]["mu"][:8], "mu_step", out[0]["mu_step"][:8], "it", out[0]["iterations"][:8], "alpha", out[0]["alpha"][:8], "impr", out[0]["improved"][:8])
for f in ("mu", "mu_step", "iterations", "alpha", "improved", "traj_costs", "termination_condition", "lg_iterations"):
    a, b = outs[0][0][f], outs[2][0][f]
    d = np.nonzero(~((a == b) | (np.isnan(a) & np.isnan(b))))[0]
    print(f, "differs at", d[:10], "single", a[d[:10]], "fused", b[d[:10]])
i = np.nonzero(outs[0][0]["mu"] != outs[2][0]["mu"])[0]
for j in i[:3]:
    print("problem", j, {f: (outs[0][0][f][j], outs[2][0][f][j]) for f in ("mu", "mu_step", "iterations", "alpha", "improved", "traj_costs", "termination_condition", "lg_iterations")})
