#!/bin/bash
# Same-box A/B of the sweep's workgroup shape (TPLHIP_NG = row pairs per workgroup) on the headline bench.
#   gpurun -- 'bash tools/ab_ng.sh [rounds] [grid]'
for r in $(seq 1 "${1:-2}"); do
  for ng in 1 2 4; do
    printf "NG=%d " $ng
    TPLHIP_NG=$ng python bench.py --steps 20 --warmup 3 --grid "${2:-target}" --no-cpu-baseline --no-secondary 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['launch_us'],2), 'us/sweep', round(d['value']), 'sweeps/s')"
  done
done
