#!/bin/bash
# The CPU test-suite against a build of libtplhip.so whose HOST code is compiled with AddressSanitizer +
# UndefinedBehaviorSanitizer (device code untouched: -fno-gpu-sanitize).  Everything the library does without a GPU --
# parameter handling, error paths, trajectory helpers, the planning host routines under the planner and controller
# plugins -- runs under the sanitizers.  CPU only; the shipped library is put back afterwards.
#   bash tools/sanitize_cpu_suite.sh
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TPL_SAN_DIR:-/tmp/tpl_san}
mkdir -p "$OUT"
cd "$REPO/tpl_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-omit-frame-pointer -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-slp-vectorize -I"$REPO/include" \
    -o "$OUT/libtplhip_san.so" $(sed -n 's/^SRCS := //p' Makefile) \
    > "$OUT/build.log" 2>&1 || { echo "sanitized build failed, see $OUT/build.log"; exit 1; }
cd "$REPO"
cp tpl_amd/libtplhip.so "$OUT/libtplhip_shipped.so"
trap 'cp "$OUT/libtplhip_shipped.so" "$REPO/tpl_amd/libtplhip.so"' EXIT
cp "$OUT/libtplhip_san.so" tpl_amd/libtplhip.so
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 python -m pytest tests -q -m "not gpu" -p no:cacheprovider

