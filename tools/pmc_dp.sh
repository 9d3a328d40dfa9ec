#!/bin/bash
# PMC passes for the DP sweep kernel (separate --pmc runs, no tracing domains).
# usage (on the GPU box, from the repo root): bash tools/pmc_dp.sh <outdir> [grid | mc] [blend order]
#   grid: target / config2 / config4 / default (tools/dp_only.py);  mc: the Monte-Carlo scenario workload (tools/mc_only.py)
#   blend order 1 | 2 | 3: the plan runs k_backward3o (set_blend_order)
set -u
OUT=${1:-gpurun_out/pmc}
GRID=${2:-target}
ORDER=${3:-0}
if [ "$GRID" = mc ]; then DRIVER="tools/mc_only.py"; DARGS="128"; else DRIVER="tools/dp_only.py"; DARGS="$GRID 1 $ORDER"; fi
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
  "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
  ; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$REPO/$OUT/pass$i" -- python3 "$REPO/$DRIVER" $DARGS > "$REPO/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" k_backward
