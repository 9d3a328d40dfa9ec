#!/bin/bash
# Same-box A/B of libtplhip.so revisions on the headline bench (see tools/ab_ilqr.sh for why).
#   here:        bash tools/ab_dp.sh build <rev|WORK> [<rev> ...] [-- extra hipcc flags]
#   on the box:  gpurun -- 'bash tools/ab_dp.sh run [rounds] [grid]'
set -eu
cd "$(dirname "$0")/.."
SRCS="common.hip dp_lon.hip dp_sweep.hip dp_sweep_rules_a.hip dp_sweep_rules_b.hip dp_sweep_rules_c.hip dp_sweep_rules_d.hip dp_sweep_rules_e.hip dp_sweep_rules_f.hip dp_sweep_rules_g.hip dp_sweep_rules_h.hip dp_sweep_rules_i.hip dp_env.hip poly_lat.hip lat_lon.hip planning.hip"
case "${1:-}" in
build)
    shift
    revs=(); extra=()
    while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; revs+=("$1"); shift; done
    mkdir -p tools/ab_tmp
    for rev in "${revs[@]}"; do
        top=$(mktemp -d); d=$top/tpl_amd/csrc
        mkdir -p $d; ln -s "$PWD/include" $top/include           # the sources include ../../include/tplhip.h
        for f in $SRCS dp_lon_math.h dp_env_internal.h dp_lon_internal.h dp_lon_kernels.h tplhip_common.h tplhip_debug.h; do
            if [ "$rev" = WORK ]; then cp tpl_amd/csrc/$f $d/; else git show "$rev:tpl_amd/csrc/$f" > $d/$f 2>/dev/null || { rm -f $d/$f; continue; }; fi
        done
        # file by file with the Makefile's flags (dp_sweep.hip without the SLP vectoriser: packed fp32 forms are slow here)
        (cd $d && for f in $SRCS; do
            [ -f $f ] || continue
            fl=""; case "$f" in dp_sweep*) fl="-fno-slp-vectorize";; esac
            /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
                -fvisibility=hidden -I"$OLDPWD/include" $fl "${extra[@]}" -c $f -o ${f%.hip}.o || exit 1
         done && ls *.o > /dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OLDPWD/tools/ab_tmp/tplhip_$rev.so" *.o) \
            2> "tools/ab_tmp/tplhip_$rev.log" || { echo "$rev: build failed"; exit 1; }
        rm -rf $top
        echo "built tools/ab_tmp/tplhip_$rev.so"
    done ;;
run)
    # (every build is loaded from its own path through TPLHIP_LIB: the shipped library is never overwritten)
    for r in $(seq 1 "${2:-2}"); do
        for so in tools/ab_tmp/tplhip_*.so; do
            printf "%-22s " "$(basename "$so" .so)"
            TPLHIP_LIB="$PWD/$so" python bench.py --steps 20 --warmup 3 --grid "${3:-target}" --no-cpu-baseline --no-secondary 2>/dev/null \
              | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['launch_us'],2), 'us/sweep', round(d['value']), 'sweeps/s')"
        done
    done ;;
*)
    echo "usage: $0 build <rev|WORK>... [-- flags] | run [rounds] [grid]"; exit 2 ;;
esac
