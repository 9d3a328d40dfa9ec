#!/bin/bash
# A third soak, weighted towards what round 5 added (GPU box, repo root):  bash tools/parity_soak_seeds3.sh [seed base]
B=${1:-900000}
run() { echo "  $*"; ( "$@" 2>&1 | tail -2 | sed 's/^/        /' ); }
runl() { echo "  $*"; ( "$@" 2>&1 | tail -14 | sed 's/^/        /' ); }
echo "third soak, seed base $B"
run python tests/fuzz_dp.py 1500 $((B + 0)) --rules --orders
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 1000 $((B + 2000)) --rules --orders
TPLHIP_DEBUG=skew=1,big=1 run python tests/fuzz_dp.py 500 $((B + 4000)) --rules --orders
TPLHIP_DEBUG=no_v3,no_s3 run python tests/fuzz_dp.py 300 $((B + 5000)) --rules --orders
TPLHIP_DEBUG=no_fastdiv run python tests/fuzz_dp.py 300 $((B + 6000)) --rules --orders
run python tests/fuzz_scenarios.py 100 $((B % 100000 + 7000))
run python tests/fuzz_engines.py 300 $((B % 100000 + 8000)) --rules
runl python tests/fuzz_ilqr.py 30
TPLHIP_DEBUG=ilqr_split=3 runl python tests/fuzz_ilqr.py 10
run python tests/fuzz_ilqr_forms.py 30
