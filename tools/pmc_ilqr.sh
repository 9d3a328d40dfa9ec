#!/bin/bash
# PMC passes for the batched iLQR kernel (SURVEY 8d asks for fp64 VALU-busy and MFMA-busy next to iterations/s).
# usage (GPU box, repo root): bash tools/pmc_ilqr.sh <outdir> [mpc|lateral] [batch] [kernel substring]
# (TPLHIP_DEBUG=ilqr_split=1 in the environment selects the split-launch form)
set -u
OUT=${1:-gpurun_out/pmc_ilqr}
WL=${2:-mpc}
BATCH=${3:-1024}
KERNEL=${4:-k_ilqr_update}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for C in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
  ; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$REPO/$OUT/pass$i" -- python3 "$REPO/tools/ilqr_only.py" "$WL" "$BATCH" > "$REPO/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" "$KERNEL" all
