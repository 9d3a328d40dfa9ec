#!/bin/bash
# Where the single wavefront of a one-problem solve spends its cycles (separate --pmc passes, no trace domains).
#   on the GPU box: bash tools/pmc_ilqr_single.sh gpurun_out/pmc_single
set -u
OUT=${1:-gpurun_out/pmc_single}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
  ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d "$REPO/$OUT/pass$i" -- python3 "$REPO/tools/ilqr_single.py" > "$REPO/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" k_ilqr_update
