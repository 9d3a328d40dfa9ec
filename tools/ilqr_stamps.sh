#!/bin/bash
# Phase breakdown of k_ilqr_update on the bench workload (1024 problems, N = 50, 20 iterations):
# builds the solver with -DTPL_ILQR_STAMP (s_memtime stamps around every phase, summed per problem)
# into gpurun_out/ and runs one update.  Diagnostic only; run on the GPU box from the repo root:
#   bash tools/ilqr_stamps.sh [batch]
set -eu
REPO=$(pwd)
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -fno-fast-math \
    -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -DTPL_ILQR_STAMP -DTPL_MODEL_INC="\"$REPO/tpl_amd/models/trajectory_tracking_mpc.inc\"" \
    -I include -o gpurun_out/libilqr_stamp.so tpl_amd/csrc/ilqr/ilqr_model.hip 2> gpurun_out/ilqr_stamp_build.log
python tools/ilqr_stamp_run.py "${1:-1024}" 2>&1 | tail -18
