"""One update of the -DTPL_ILQR_STAMP build of the MPC solver (tools/ilqr_stamps.sh): prints the phase stamps."""
import sys, os, numpy as np
sys.path.insert(0, ".")
from tpl_amd.optim import bench_ilqr, genopt, optimizers as o, solver
cfg = o.config_trajectory_tracking_mpc()
make = solver.factory(os.path.abspath("gpurun_out/libilqr_stamp.so"), cfg.param_defaults)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
opt = make(batch=batch, horizon=50, device=0)
x0 = bench_ilqr.initial_states(batch)
bench_ilqr._configure(opt, x0 if batch > 1 else x0[0])
opt.upload()
opt.restore(); opt.run_resident(); opt.sync()
opt.restore(); opt.run_resident(); opt.sync()
opt.download()
print("iterations", np.asarray(opt.iterations).reshape(-1)[0])
