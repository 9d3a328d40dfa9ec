#!/usr/bin/env python3
"""Randomised parity soak of the HIP iLQR solver against the CPU restatement (oracle/ilqr_oracle.cpp): every tpl problem
over many seeds (start states, curvatures, horizons), two consecutive update() calls with a shift in between.  Reports
the worst scaled trajectory error and every case whose iteration count or line-search outcome differs (a knife-edge
accept/reject decision taken differently because device and host transcendentals differ in the last bit).

    python tests/fuzz_ilqr.py [seeds per problem]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ilqr_cases as cases                      # noqa: E402
import oracle_ilqr as OI                        # noqa: E402
from tpl_amd.optim import genopt, optimizers as o   # noqa: E402

PLANS = [
    ("trajectory_tracking_mpc", lambda opt, s: cases.mpc(opt, seed=s, curvature=[0.0, 0.02, -0.03, 0.05][s % 4],
                                                          horizon=[20, 50, 60, 35][s % 4], min_rel=[1e-6, 0.0][s % 2])),
    ("trajectory_tracking_mpc_time", lambda opt, s: cases.mpc_time(opt, seed=s)),
    ("lateral_profile", lambda opt, s: cases.lateral(opt, seed=s, horizon=[120, 250, 64][s % 3])),
    ("velocity_profile_space", lambda opt, s: cases.velocity_space(opt, seed=s, horizon=[100, 180][s % 2])),
    ("ref_line_smoother_k", lambda opt, s: cases.smoother(opt, seed=s, horizon=[80, 150][s % 2])),
    ("ref_line_smoother_dk", lambda opt, s: cases.smoother(opt, seed=s, with_dk=True)),
    ("velocity_profile_time", lambda opt, s: cases.velocity_time(opt, seed=s)),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    worst, odd = {}, []
    for model, setup in PLANS:
        make = genopt.build(getattr(o, "config_" + model)(), model)
        for s in range(n):
            hip, ref = make(), OI.OracleOptim(model)
            for opt in (hip, ref):
                setup(opt, s)
            err = 0.0
            for step in range(2):
                if step:
                    for opt in (hip, ref):
                        opt.shift(1 + s % 3)
                hip.update()
                ref.update()
                same = (int(hip.iterations) == int(ref.iterations) and float(hip.alpha) == float(ref.alpha)
                        and bool(hip.improved) == bool(ref.improved))
                e = max(cases.scaled_err(hip.x, ref.x), cases.scaled_err(hip.u, ref.u))
                if not same:
                    odd.append((model, s, step, int(hip.iterations), int(ref.iterations), float(hip.alpha), float(ref.alpha), e))
                    break                                    # from here on the two solvers solve different problems
                err = max(err, e)
            worst[model] = max(worst.get(model, 0.0), err)
    for m, e in worst.items():
        print(f"{m:32s} worst scaled error {e:.2e}")
    print(f"{len(odd)} of {n * len(PLANS)} cases took a different accept/reject path:")
    for row in odd:
        print("   ", row)


if __name__ == "__main__":
    main()
