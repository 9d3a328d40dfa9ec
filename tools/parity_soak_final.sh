#!/bin/bash
# The soak of the round's final libraries (GPU box, repo root): bash tools/parity_soak_final.sh [seed base]
B=${1:-1700000}
run() { echo "  $*"; ( "$@" 2>&1 | tail -2 | sed 's/^/        /' ); }
runl() { echo "  $*"; ( "$@" 2>&1 | tail -12 | sed 's/^/        /' ); }
echo "final soak, seed base $B, $(git log -1 --format=%h 2>/dev/null || echo snapshot)"
run python tests/fuzz_dp.py 500 $((B + 0)) --rules --orders
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 700 $((B + 2000)) --rules --orders
TPLHIP_DEBUG=skew=1,big=1 run python tests/fuzz_dp.py 200 $((B + 4000)) --rules --orders
TPLHIP_DEBUG=no_fastdiv,big=1 run python tests/fuzz_dp.py 150 $((B + 6000)) --rules --orders
run python tests/fuzz_scenarios.py 60 $((B % 100000 + 7000))
run python tests/fuzz_engines.py 100 $((B % 100000 + 8000)) --rules
runl python tests/fuzz_ilqr.py 30
TPLHIP_DEBUG=ilqr_split=2 runl python tests/fuzz_ilqr.py 10
run python tests/fuzz_ilqr_forms.py 30
