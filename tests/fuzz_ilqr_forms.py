#!/usr/bin/env python3
"""Randomised bit-identity soak of the launch forms of the batched iLQR update() (single launch / split sequence /
fused launch / the fused phases at one wavefront per SIMD, TPLHIP_DEBUG=ilqr_split=0|1|2|3): every problem of the goldens as a ragged batch -- per-problem start states,
horizons and iteration budgets, a random integrator (EULER / HEUN / RK4: template parameters of the chain phase), second- or
first-order method, one or two multiplier rounds, two updates with a shift in between.  Every field and scalar must agree
bit for bit between the forms; the oracle is not involved (tests/fuzz_ilqr.py compares the solver with it).

    python tests/fuzz_ilqr_forms.py [seeds per problem]          (TPL_FORMS_FUZZ_DEBUG=ilqr_no_lds adds debug tokens to every leg)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ilqr_cases as cases                      # noqa: E402
from tpl_amd.optim import genopt, optimizers as o   # noqa: E402

GOLDEN = os.path.join(HERE, "golden")
FIELDS = ("x", "u", "k", "K", "lagrange_multiplier", "next_x", "next_u", "prev_x", "prev_k", "fx", "fu", "lx", "lu",
          "lxx", "luu", "lux", "traj_costs", "iterations", "lg_iterations", "alpha", "mu", "mu_step",
          "trajectory_changed", "improved", "termination_condition")
FORMS = {0: "single", 1: "split", 2: "fused", 3: "fused1"}
EXTRA = os.environ.get("TPL_FORMS_FUZZ_DEBUG", "")     # further TPLHIP_DEBUG tokens for every leg (ilqr_no_lds, ilqr_split_lds_kb=8 ...)


def run(model, golden, seed, form):
    os.environ["TPLHIP_DEBUG"] = f"ilqr_split={form}" + ("," + EXTRA if EXTRA else "")
    rng = np.random.default_rng(seed)
    make = genopt.build(getattr(o, "config_" + model)(), model)
    g = np.load(os.path.join(GOLDEN, f"ilqr_{golden}.npz"))
    one = make()
    cases.apply_inputs(one, g)
    T = int(one.horizon)
    B = int(rng.integers(3, 41))
    opt = make(batch=B, horizon=T)
    cases.apply_inputs(opt, g)
    x0 = np.array(one.x[0])
    for b in range(B):
        opt.x[b, 0] = x0 * (1.0 + 0.1 * rng.uniform(-1, 1, x0.shape)) + 0.01 * rng.uniform(-1, 1, x0.shape)
    integ, second = int(rng.integers(0, 3)), bool(rng.integers(0, 4))   # mostly the second-order method
    opt.integrator_type = integ
    opt.use_quadratic_terms = second
    opt.max_lg_iterations = int(rng.integers(1, 3))
    opt.max_iterations = rng.integers(1, 7, B)
    opt.horizon = T - rng.integers(0, max(1, T // 3), B)
    opt.min_rel_cost_change = [0.0, 1e-3, 1e-6][int(rng.integers(0, 3))]
    out = []
    for step in range(2):
        opt.update()
        out.append({f: np.array(getattr(opt, f)) for f in FIELDS})
        opt.shift(1 + seed % 3)
    info = opt.info()
    assert info["form"] == FORMS[form], (model, seed, info)
    return out, (B, integ, second)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    problems = sorted({(c[0], name) for name, c in cases.CASES.items()})
    seen, bad, total = set(), [], 0
    for model, golden in problems:
        for s in range(n):
            ref, what = run(model, golden, 1000 * s + 7, 0)
            for form in (1, 2, 3):
                got, _ = run(model, golden, 1000 * s + 7, form)
                for step, (a, b) in enumerate(zip(ref, got)):
                    for f in FIELDS:
                        if not np.array_equal(a[f], b[f], equal_nan=True):
                            bad.append((model, golden, s, FORMS[form], step, f))
            seen.add((model,) + what[1:])
            total += 1
    print(f"{total} ragged batches x 4 forms x 2 updates over {len({p[0] for p in problems})} models; "
          f"(model, integrator, second order) combinations drawn: {len(seen)}")
    print(f"{len(bad)} field mismatches between the forms" + (":" if bad else " -- all four forms bit-identical"))
    for row in bad[:40]:
        print("   ", row)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
