#!/bin/bash
# Evidence for the batched iLQR after a solver change (GPU box, repo root): kernel traces and PMC passes of the single launch
# (1 024 problems) and of the fused form (4 096), the three forms over the batch size -> gpurun_out/r05i
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05i; mkdir -p "$OUT"
(cd /tmp && export TMPDIR=/tmp PYTHONPATH="$REPO" && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr" -- python3 "$REPO/tools/ilqr_only.py" mpc > "$OUT/trace_ilqr.log" 2>&1)
(cd /tmp && export TMPDIR=/tmp PYTHONPATH="$REPO" && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr_fused" -- python3 "$REPO/tools/ilqr_only.py" mpc 4096 > "$OUT/trace_ilqr_fused.log" 2>&1)
bash tools/pmc_ilqr.sh gpurun_out/r05i/pmc_ilqr mpc > "$OUT/pmc_ilqr.txt" 2>&1
bash tools/pmc_ilqr.sh gpurun_out/r05i/pmc_ilqr_fused mpc 4096 k_ilqr_fused > "$OUT/pmc_ilqr_fused.txt" 2>&1
python3 tools/ab_ilqr_split.py mpc --forms 0,1,2 1024 1536 2048 4096 8192 16384 > "$OUT/ilqr_forms_ab.txt" 2>&1
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
tail -20 "$OUT/ilqr_forms_ab.txt" | cut -c1-200
