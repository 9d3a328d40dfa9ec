#!/bin/bash
# Traffic and time of k_backward3v with and without the skewed traversal (TPLHIP_DEBUG=skew=0|1), per grid:
# FETCH_SIZE / WRITE_SIZE / L2 hits from separate --pmc passes (plus the DRAM-destined request counters), launch time from
# the un-profiled bench.      on the GPU box: bash tools/skew_traffic_study.sh gpurun_out/r04/skew "config4 target"
set -u
OUT=${1:-gpurun_out/skew}
GRIDS=${2:-"config4 target"}
EXTRA=${3:-}
REPO=$(pwd)
mkdir -p "$OUT"
for G in $GRIDS; do
  for K in 0 1; do
    export TPLHIP_DEBUG="skew=$K${EXTRA:+,$EXTRA}"
    T=$(timeout 200 python bench.py --grid $G --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null \
        | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['roofline']['launch_us'],2))")
    D=$OUT/${G}_skew$K
    (cd /tmp && export TMPDIR=/tmp && \
     timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$REPO/$D/pass1" -- python3 "$REPO/tools/dp_only.py" $G 1 > "$REPO/$D.p1.log" 2>&1; \
     timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$REPO/$D/pass2" -- python3 "$REPO/tools/dp_only.py" $G 1 > "$REPO/$D.p2.log" 2>&1; \
     timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum --output-format csv -d "$REPO/$D/pass3" -- python3 "$REPO/tools/dp_only.py" $G 1 > "$REPO/$D.p3.log" 2>&1; \
     timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d "$REPO/$D/pass4" -- python3 "$REPO/tools/dp_only.py" $G 1 > "$REPO/$D.p4.log" 2>&1)
    echo "== $G skew=$K launch_us $T"
    python3 tools/pmc_summary.py "$D" k_backward3v 2>/dev/null | grep -v "^#"
    find "$D" -name "*.csv" -delete
  done
done
