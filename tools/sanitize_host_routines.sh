#!/bin/bash
# Host-only ASan + UBSan build of the planning host routines and a random-input run through them (CPU, no GPU needed).
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TPL_SAN_DIR:-/tmp/tpl_san}
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-host-only -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -I"$REPO/include" -o "$OUT/libplan_san.so" "$REPO/tpl_amd/csrc/planning.hip" "$REPO/tpl_amd/csrc/common.hip"
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
TPL_SAN_LIB="$OUT/libplan_san.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0 python3 "$REPO/tools/sanitize_host_routines.py"
