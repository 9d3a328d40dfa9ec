#!/bin/bash
# Phase breakdown of ONE single-problem solve of a small problem (default: lateral_profile, T = 250) -- what tplsim's
# default planner calls twice per cycle.  Diagnostic build with s_memtime stamps; on the GPU box, from the repo root:
#   bash tools/ilqr_stamps_small.sh [lateral_profile|velocity_profile_space]
set -eu
REPO=$(pwd)
M=${1:-lateral_profile}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -fno-fast-math \
    -ffp-contract=off -DTPL_ILQR_STAMP -DTPL_MODEL_INC="\"$REPO/tpl_amd/models/$M.inc\"" \
    -I include -o gpurun_out/libilqr_stamp_$M.so tpl_amd/csrc/ilqr/ilqr_model.hip 2> gpurun_out/ilqr_stamp_build_$M.log
python - "$M" <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ilqr_cases as cases
from tpl_amd.optim import optimizers as o, solver
m = sys.argv[1]
cfg = getattr(o, "config_" + m)()
make = solver.factory(os.path.abspath(f"gpurun_out/libilqr_stamp_{m}.so"), cfg.param_defaults)
opt = make(device=0)
(cases.lateral if m == "lateral_profile" else cases.velocity_space)(opt, seed=0, horizon=250)
opt.min_rel_cost_change = 0.0
opt.upload(); opt.run_resident(); opt.sync(); opt.download()
print("iterations", int(np.ravel(opt.iterations)[0]))
PY
