#!/bin/bash
# Same-box A/B of libtplhip.so builds (tools/ab_dp.sh build ...) on the SMALL grids: default-grid plan latency and the
# 128x64x32 sweep.   gpurun -- 'bash tools/ab_small.sh [rounds]'
for r in $(seq 1 "${1:-2}"); do
  for so in tools/ab_tmp/tplhip_*.so; do
    export TPLHIP_LIB="$PWD/$so"      # loaded from its own path: the shipped library is never overwritten
    printf "%-20s default-grid plan p50/p99 ms: %s   " "$(basename "$so" .so)" "$(python tools/dp_default_plan.py)"
    python bench.py --steps 20 --warmup 3 --grid config2 --no-cpu-baseline --no-secondary 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('128x64x32:', round(d['roofline']['launch_us'],2), 'us/sweep')"
  done
done
unset TPLHIP_LIB
