#!/usr/bin/env python3
"""Wall time of tplsim's default planner plugin (PathVelDecompPlanner) per cycle on the HIP solvers, and of its stages."""
import os
import sys
import time
from contextlib import contextmanager

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pvd_cases as cases                                                   # noqa: E402
from tpl_amd.planning.path_vel_decomp import PathVelDecompPlanner           # noqa: E402
from tpl_amd.store import Bundle                                            # noqa: E402


@contextmanager
def lock():
    yield


def main():
    solvers = None
    if len(sys.argv) > 1 and sys.argv[1] == "cpu":
        import oracle_ilqr as OI
        solvers = {k: OI.OracleOptim(k) for k in ("ref_line_smoother_k", "lateral_profile", "velocity_profile_space")}
    sh = Bundle()
    pl = PathVelDecompPlanner(sh, lock, solvers=solvers)
    for kind in ("path_optim", "velocity_optim", "path_smoothing"):
        for k, v in vars(cases.default_params(kind)).items():
            setattr(getattr(sh.params, kind), k, v)
    total, lat, vel = [], [], []
    for cyc in range(40):
        env = cases.shared_environment(cyc % 12)
        t0 = time.perf_counter()
        pl.update(env)
        total.append((time.perf_counter() - t0) * 1e3)
        lat.append(pl.path_optim.update.runtime * 1e3)
        vel.append(pl.velocity_optim.update.runtime * 1e3)
    f = lambda a: round(float(np.percentile(a[5:], 50)), 3)
    print("PathVelDecompPlanner.update p50 ms:", f(total), " lateral stage:", f(lat), " speed stage:", f(vel),
          " iterations:", int(np.ravel(pl.path_optim.opt.iterations)[0]), int(np.ravel(pl.velocity_optim.opt.iterations)[0]))


if __name__ == "__main__":
    main()
