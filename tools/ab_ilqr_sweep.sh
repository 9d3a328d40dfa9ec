#!/bin/bash
# Same-box batch sweep of the iLQR solver builds in tools/ab_tmp/*.so (see tools/ab_ilqr.sh for how they get there):
#   gpurun -- 'bash tools/ab_ilqr_sweep.sh [rounds]'
# Every build is loaded from its own path (tools/ilqr_only.py takes it); the shipped solver is never overwritten.
# (tools/ab_ilqr_split.py does the same per launch form: single launch / split sequence.)
set -eu
cd "$(dirname "$0")/.."
for r in $(seq 1 "${1:-1}"); do
    for so in tools/ab_tmp/*.so; do
        printf "%-14s " "$(basename "$so" .so)"
        for b in 256 512 1024 2048 4096; do
            printf "%d: %s   " $b "$(python3 tools/ilqr_only.py mpc $b "$PWD/$so" 2>&1 | tail -1)"
        done
        echo
    done
done
