#!/bin/bash
# FETCH_SIZE / L2 hit rate of the DP sweep over traversal tile shapes (TPLHIP_TILE_V x TPLHIP_TILE_A).
# usage (GPU box, repo root): bash tools/pmc_fetch_sweep.sh <outdir> [grid]
set -u
OUT=${1:-gpurun_out/pmc_fetch}
GRID=${2:-config4}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for TV in 2 4 8; do for TA in 2 4 8 16; do
  export TPLHIP_TILE_V=$TV TPLHIP_TILE_A=$TA
  D="$REPO/$OUT/v${TV}_a${TA}"
  mkdir -p "$D"
  rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$D/pass1" -- python3 "$REPO/tools/dp_only.py" "$GRID" 1 > "$D/pass1.log" 2>&1 || echo "failed $TV $TA"
  echo "tile_v=$TV tile_a=$TA"; python3 "$REPO/tools/pmc_summary.py" "$D" k_backward | grep -E "FETCH|TCC"
  find "$D" -name "*.csv" -delete
done; done
