#!/bin/bash
# Kernel-trace A/B of libtplhip.so builds: the forward rollout of a 256x128x64 x 50 plan, not overlapped with sweeps.
R=$PWD
for so in tools/ab_tmp/tplhip_*.so; do
  export TPLHIP_LIB="$R/$so"      # loaded from its own path: the shipped library is never overwritten
  n=$(basename "$so" .so)
  (cd /tmp && TMPDIR=/tmp TPLHIP_DEBUG=no_fwd_overlap${AB_EXTRA:+,$AB_EXTRA} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abfwd_$n -- python3 $R/tools/dp_only.py ${1:-target} 3 > /dev/null 2>&1)
  python3 - $n <<'PY'
import csv,glob,sys
f=sorted(glob.glob(f'gpurun_out/abfwd_{sys.argv[1]}/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'k_forward' in r['Name']:
        print(sys.argv[1], r['Calls'], f"{float(r['AverageNs'])/1e3:.2f} us (min {float(r['MinNs'])/1e3:.2f})")
PY
done
unset TPLHIP_LIB
