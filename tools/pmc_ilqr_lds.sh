#!/bin/bash
# Where the LDS bank conflicts of the batched iLQR kernel come from: the same PMC pass on the shipped solver and on the
# two ablation builds (TPL_ILQR_ABLATE=1: no backward pass, =2: no rollouts; results are not solutions).
#   here:        bash tools/pmc_ilqr_lds.sh build
#   on the box:  gpurun -- 'bash tools/pmc_ilqr_lds.sh run gpurun_out/pmc_ilqr_lds'
set -eu
cd "$(dirname "$0")/.."
REPO=$PWD
INC="$REPO/tpl_amd/models/trajectory_tracking_mpc.inc"
case "${1:-}" in
build)
    mkdir -p tools/ab_tmp
    for v in 0 1 2; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -fno-fast-math \
            -ffp-contract=off -DTPL_ILQR_ABLATE=$v -DTPL_MODEL_INC="\"$INC\"" -I include -o tools/ab_tmp/ilqr_abl$v.so \
            tpl_amd/csrc/ilqr/ilqr_model.hip
        echo "built tools/ab_tmp/ilqr_abl$v.so"
    done ;;
run)
    OUT=${2:-gpurun_out/pmc_ilqr_lds}
    mkdir -p "$OUT"
    # (the ablation builds are loaded from their own path -- tools/ilqr_only.py takes it -- and never installed over the
    # shipped solver: they do not produce solutions)
    for v in 0 1 2; do
        (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES \
            --output-format csv -d "$REPO/$OUT/abl$v/pass1" -- python3 "$REPO/tools/ilqr_only.py" mpc 1024 "$REPO/tools/ab_tmp/ilqr_abl$v.so" > "$REPO/$OUT/abl$v.log" 2>&1) || echo "abl$v failed"
        echo "== TPL_ILQR_ABLATE=$v"; python3 tools/pmc_summary.py "$OUT/abl$v" k_ilqr_update
    done ;;
*) echo "usage: $0 build | run [outdir]"; exit 2 ;;
esac
