"""p50 wall time of Optim.update() in the closed MPC loop of benchkit/latency.py, with the kernel share."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from tpl_amd.optim import genopt, optimizers as o
from benchkit import ilqr as bench_ilqr
make = genopt.build(o.config_trajectory_tracking_mpc(), "trajectory_tracking_mpc")
opt = make(device=0)
opt.horizon = 20
x = np.array([0.0, 0.4, 0.05, 0.0, 6.0, 0.0, 0.0])
bench_ilqr._configure(opt, x)
opt.min_rel_cost_change = 1e-6
opt.horizon = 20
lat = []
opt.timing(reset=True)
for k in range(200):
    opt.x[0] = x
    t0 = time.perf_counter()
    opt.update()
    lat.append((time.perf_counter() - t0) * 1e3)
    x = opt.dynamics(x, opt.u[0], 0, 0.01)
    x[5] = max(0.0, x[0])
    opt.shift(0)
km, n = opt.timing(reset=True)
print("p50 total ms", np.percentile(lat[10:], 50), "kernel ms/update", km / max(1, n), "launches", n, "iters", opt.iterations)
