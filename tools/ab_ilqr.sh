#!/bin/bash
# Same-box A/B of iLQR solver revisions (launch times differ by +-2 % between the boxes of the
# pool, so two builds are only comparable inside ONE gpurun call).
#   here:        bash tools/ab_ilqr.sh build <rev> [<rev> ...]     (rev: a git revision or HEAD/WORK for the tree)
#   on the box:  gpurun -- 'bash tools/ab_ilqr.sh run [rounds]'
# Builds land in tools/ab_tmp/ (git-ignored, travels with the snapshot).
set -eu
cd "$(dirname "$0")/.."
INC="$PWD/tpl_amd/models/trajectory_tracking_mpc.inc"
case "${1:-}" in
build)
    shift
    mkdir -p tools/ab_tmp
    for rev in "$@"; do
        src=/tmp/ab_ilqr_$$.hip
        if [ "$rev" = WORK ]; then cp tpl_amd/csrc/ilqr/ilqr_model.hip $src; else git show "$rev:tpl_amd/csrc/ilqr/ilqr_model.hip" > $src; fi
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -fno-fast-math \
            -ffp-contract=off -DTPL_MODEL_INC="\"$INC\"" -I include -I tpl_amd/csrc/ilqr \
            -o "tools/ab_tmp/$rev.so" -x hip $src 2> "tools/ab_tmp/$rev.log" || { echo "$rev: build failed"; exit 1; }
        rm -f $src
        echo "built tools/ab_tmp/$rev.so (headers and model text of the working tree)"
    done ;;
run)
    # (every build is loaded from its own path; the shipped solver is never overwritten)
    for r in $(seq 1 "${2:-2}"); do
        for so in tools/ab_tmp/*.so; do
            printf "%-14s " "$(basename "$so" .so)"
            python3 tools/ilqr_only.py mpc 1024 "$PWD/$so" 2>&1 | tail -1
        done
    done ;;
*)
    echo "usage: $0 build <rev>... | run [rounds]"; exit 2 ;;
esac
