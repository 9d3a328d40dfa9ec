#!/bin/bash
# Traffic and time of k_backward3v on 512x256x128 per traversal tile shape (TPLHIP_DEBUG=tile_v=..,tile_a=..):
# FETCH_SIZE / WRITE_SIZE / L2 hit rate from two separate --pmc passes, launch time from the un-profiled bench.
#   on the GPU box: bash tools/tile_traffic_study.sh gpurun_out/tiles "4x8 8x8 2x8 4x16 4x4 8x4 4x32 2x32 16x4 8x16"
set -u
OUT=${1:-gpurun_out/tiles}
SHAPES=${2:-"4x8 8x8 2x8 4x16"}
REPO=$(pwd)
mkdir -p "$OUT"
for S in $SHAPES; do
  export TPLHIP_DEBUG="tile_v=${S%x*},tile_a=${S#*x}"
  T=$(timeout 120 python bench.py --grid config4 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null \
      | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['roofline']['launch_us'],1))")
  (cd /tmp && export TMPDIR=/tmp && \
   timeout 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$REPO/$OUT/$S/pass1" -- python3 "$REPO/tools/dp_only.py" config4 1 > "$REPO/$OUT/$S.p1.log" 2>&1; \
   timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$REPO/$OUT/$S/pass2" -- python3 "$REPO/tools/dp_only.py" config4 1 > "$REPO/$OUT/$S.p2.log" 2>&1)
  echo "== $S launch_us $T"
  python3 tools/pmc_summary.py "$OUT/$S" k_backward 2>/dev/null | grep -E "FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS"
  find "$OUT/$S" -name "*.csv" -delete
done
