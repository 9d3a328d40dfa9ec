#!/bin/bash
# Where a single-problem iteration goes, by leaving parts out (diagnostic builds, results are NOT solutions):
#   bash tools/ilqr_ablate.sh [model]      on the GPU box, from the repo root
set -eu
REPO=$(pwd)
M=${1:-lateral_profile}
mkdir -p gpurun_out
for V in "0 -DTPL_ILQR_ABLATE=0" "1 -DTPL_ILQR_ABLATE=1" "2 -DTPL_ILQR_ABLATE=2" "g -DTPL_ILQR_SMALL_NX=0"; do
  set -- $V
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -fno-fast-math \
      -ffp-contract=off $2 -DTPL_MODEL_INC="\"$REPO/tpl_amd/models/$M.inc\"" \
      -I include -o gpurun_out/libilqr_abl_$1.so tpl_amd/csrc/ilqr/ilqr_model.hip 2> gpurun_out/ilqr_abl_$1.log
done
python - "$M" <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ilqr_cases as cases
from tpl_amd.optim import optimizers as o, solver
m = sys.argv[1]
cfg = getattr(o, "config_" + m)()
setup = {"lateral_profile": lambda opt: cases.lateral(opt, seed=0, horizon=250),
         "velocity_profile_space": lambda opt: cases.velocity_space(opt, seed=0, horizon=250)}[m]
sol = {}
for tag, what in (("0", "shipped"), ("g", "lane-parallel backward pass (TPL_ILQR_SMALL_NX=0)"), ("1", "no backward pass"), ("2", "no rollouts")):
    make = solver.factory(os.path.abspath(f"gpurun_out/libilqr_abl_{tag}.so"), cfg.param_defaults)
    opt = make(device=0)
    ks = []
    for rep in range(10):
        setup(opt)
        opt.max_iterations = 16
        opt.min_rel_cost_change = 0.0
        opt.timing(reset=True)
        opt.update()
        k, n = opt.timing(reset=True)
        ks.append(k / max(1, n))
    sol[tag] = (np.array(opt.x), np.array(opt.u), np.array(opt.K))
    print(f"{what:52s} {np.median(ks[3:]):.3f} ms per update, {int(np.ravel(opt.iterations)[0])} iterations")
print("register-resident and lane-parallel backward pass give identical bits (x, u, K):",
      all(np.array_equal(a, b) for a, b in zip(sol["0"], sol["g"])))
PY
