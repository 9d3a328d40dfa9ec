#!/bin/bash
# Same-box batch sweep of the iLQR solver builds in tools/ab_tmp/*.so (see tools/ab_ilqr.sh for how they get there):
#   gpurun -- 'bash tools/ab_ilqr_sweep.sh [rounds]'
set -eu
cd "$(dirname "$0")/.."
LIB=tpl_amd/models/libilqr_trajectory_tracking_mpc.so
cp $LIB tools/ab_tmp/.shipped.so
for r in $(seq 1 "${1:-1}"); do
    for so in tools/ab_tmp/*.so; do
        cp "$so" $LIB
        printf "%-14s " "$(basename "$so" .so)"
        python3 -m benchkit.ilqr --batch-sweep 2>&1 | tail -1 | python3 -c "
import sys, json
print('  '.join('%d: %.2fM (%.2f ms)' % (r['batch'], r['problem_iterations_per_s'] / 1e6, r['kernel_ms_per_update']) for r in json.loads(sys.stdin.read())))"
    done
done
cp tools/ab_tmp/.shipped.so $LIB
