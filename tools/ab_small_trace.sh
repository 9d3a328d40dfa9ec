#!/bin/bash
# Kernel-trace A/B of libtplhip.so builds on the default-grid plan: average k_backward3s time per build.
R=$PWD
for so in tools/ab_tmp/tplhip_*.so; do
  export TPLHIP_LIB="$R/$so"      # loaded from its own path: the shipped library is never overwritten
  n=$(basename "$so" .so)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abtrace_$n -- python3 $R/tools/dp_default_plan.py > /dev/null 2>&1)
  python3 - $n <<'PY'
import csv,glob,sys
f=glob.glob(f'gpurun_out/abtrace_{sys.argv[1]}/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_backward3s' in r['Name'] or 'k_forward' in r['Name']:
        print(sys.argv[1], r['Name'].split('(')[0][-30:], r['Calls'], f"{float(r['AverageNs'])/1e3:.2f} us (min {float(r['MinNs'])/1e3:.2f})")
PY
done
unset TPLHIP_LIB
