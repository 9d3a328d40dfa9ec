#!/usr/bin/env python3
"""Run the batched MPC bench problem on a stamped diagnostic build of the solver.
usage: python3 tools/ilqr_stamp.py <path to stamped .so> [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tpl_amd.optim import solver
from benchkit import ilqr as bench_ilqr
so = sys.argv[1]; batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
opt = solver.Optim(so, batch=batch, horizon=50, device=0)
bench_ilqr._configure(opt, bench_ilqr.initial_states(batch))
opt.upload()
opt.run_resident(); opt.sync()
opt.restore(); t0 = time.perf_counter(); opt.run_resident(); opt.sync(); el = time.perf_counter() - t0
opt.download()
print(f"batch {batch}: {el*1e3:.2f} ms per update, iterations {int(np.ravel(opt.iterations)[0])}")
