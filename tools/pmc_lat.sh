#!/bin/bash
# Latency-side PMC passes for the DP sweep (in-flight levels / instruction counts = average latency in cycles).
# usage (GPU box, repo root): bash tools/pmc_lat.sh <outdir> [grid]
set -u
OUT=${1:-gpurun_out/pmc_lat}
GRID=${2:-target}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
  ; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$REPO/$OUT/pass$i" -- python3 "$REPO/tools/dp_only.py" "$GRID" 1 > "$REPO/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" k_backward
