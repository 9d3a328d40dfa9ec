#!/bin/bash
# Same-box A/B of the library builds under tools/ab_tmp/ (tplhip_<name>.so, loaded through TPLHIP_LIB): `rounds` passes over
# all of them in turn, one process each (tools/dp_time.py: the sweep alone on the chip, min / median of five timings).
#   gpurun -- 'bash tools/ab_libs.sh [rounds] [grid] [R1]'      R1 = 0: time the libraries under set_contraction(False)
set -eu
cd "$(dirname "$0")/.."
for r in $(seq 1 "${1:-3}"); do
    for so in tools/ab_tmp/tplhip_*.so; do
        TPLHIP_LIB="$PWD/$so" TPL_AB_R1="${3:-1}" python tools/dp_time.py "${2:-target}" 2>/dev/null | sed 's/{.*}//'
    done
done
