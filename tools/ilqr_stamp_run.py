"""One update of the -DTPL_ILQR_STAMP build of the MPC solver (tools/ilqr_stamps.sh): prints the phase stamps.
usage: python tools/ilqr_stamp_run.py [batch] [horizon] [closed-loop ticks]
With ticks > 0 (batch 1) the stamps are those of the LAST tick of the mini-tplsim closed loop (warm-started, a
few iterations), i.e. what one controller cycle really costs."""
import sys, os, numpy as np
sys.path.insert(0, ".")
from tpl_amd.optim import genopt, optimizers as o, solver
from benchkit import ilqr as bench_ilqr
cfg = o.config_trajectory_tracking_mpc()
make = solver.factory(os.path.abspath("gpurun_out/libilqr_stamp.so"), cfg.param_defaults)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if ticks > 0:
    from benchkit import minisim
    opt = make(device=0)
    opt.horizon = horizon
    os.environ["TPL_ILQR_STAMP_QUIET"] = "1"
    controls, trace, con = minisim.closed_loop(opt, ticks=ticks)
    opt.download()                       # prints the [stamp] lines of the last tick's kernel
    print("closed loop: p50 ms", np.percentile(con.solve_ms[10:], 50), "iterations of the last tick", opt.iterations,
          "mean iterations over the last 100 ticks: n/a (per-tick value only)")
else:
    opt = make(batch=batch, horizon=horizon, device=0)
    x0 = bench_ilqr.initial_states(batch)
    bench_ilqr._configure(opt, x0 if batch > 1 else x0[0])
    opt.upload()
    opt.restore(); opt.run_resident(); opt.sync()
    opt.restore(); opt.run_resident(); opt.sync()
    opt.download()
    print("iterations", np.asarray(opt.iterations).reshape(-1)[0])
