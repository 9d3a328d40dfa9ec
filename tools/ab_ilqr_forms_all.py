"""Same-box table of the three update() forms for EVERY tpl problem of the goldens at large batch sizes: which models gain
from the two-wavefront forms (the library picks the fused form by itself for EVERY model above 4 x CUs problems: five of the
seven gain 22-42 % there, the two reference-line smoothers are indifferent -- table at the end of profiles/r04_ilqr_split_ab.txt).
Each problem is set up from its golden's inputs, replicated with perturbed start states, min_rel_cost_change = 0 so that
every problem runs its whole iteration budget.     python tools/ab_ilqr_forms_all.py [batches...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ilqr_cases as cases  # noqa: E402
from tpl_amd.optim import genopt, optimizers as o  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PROBLEMS = [("trajectory_tracking_mpc", "mpc_straight_fixed_iters"), ("trajectory_tracking_mpc_time", "mpc_time"),
            ("lateral_profile", "lateral"), ("velocity_profile_space", "velocity_space"), ("ref_line_smoother_k", "smoother_k"),
            ("ref_line_smoother_dk", "smoother_dk"), ("velocity_profile_time", "velocity_time")]


def run(model, golden, batch, form, iters=10):
    os.environ["TPLHIP_DEBUG"] = f"ilqr_split={form}"
    make = genopt.build(getattr(o, "config_" + model)(), model)
    g = np.load(os.path.join(GOLDEN, f"ilqr_{golden}.npz"))
    one = make()
    cases.apply_inputs(one, g)
    T = int(one.horizon)
    opt = make(batch=batch, horizon=T)
    cases.apply_inputs(opt, g)
    rng = np.random.default_rng(11)
    x0 = np.array(one.x[0])
    opt.x[:, 0] = x0[None, :] * (1.0 + 0.05 * rng.uniform(-1, 1, (batch, x0.size))) + 0.01 * rng.uniform(-1, 1, (batch, x0.size))
    opt.max_iterations = iters
    opt.max_lg_iterations = 1
    opt.min_rel_cost_change = 0.0
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
    it = float(np.mean(opt.iterations))
    return {"model": model, "X": int(opt.x.shape[-1]), "T": T, "batch": batch, "form": opt.info()["form"],
            "problem_iterations_per_s": round(3 * it * batch / el), "kernel_ms_per_update": round(ms / max(1, n), 3),
            "finite": bool(np.all(np.isfinite(opt.traj_costs)))}


if __name__ == "__main__":
    batches = [int(a) for a in sys.argv[1:]] or [2048, 8192]
    for model, golden in PROBLEMS:
        for b in batches:
            for form in (0, 2):
                print(json.dumps(run(model, golden, b, form)), flush=True)
