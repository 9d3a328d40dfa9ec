#!/bin/bash
# A second randomised parity soak with seeds of its own (GPU box, repo root):  bash tools/parity_soak_seeds2.sh [seed base] > gpurun_out/<tag>/parity_soak_seeds2.txt
B=${1:-170000}
run() { echo "  $*"; ( "$@" 2>&1 | tail -2 | sed 's/^/        /' ); }
echo "second soak, seed base $B"
run python tests/fuzz_dp.py 1500 $((B + 0))
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 600 $((B + 1000))
run python tests/fuzz_dp.py 500 $((B + 2000)) --rules
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 250 $((B + 3000)) --rules
TPLHIP_DEBUG=skew=1,big=1 run python tests/fuzz_dp.py 300 $((B + 4000))
run python tests/fuzz_dp.py 1000 $((B + 5000)) --rules --orders
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 600 $((B + 6000)) --rules --orders
run python tests/fuzz_scenarios.py 100 1500
run python tests/fuzz_engines.py 80 1200
run python tests/fuzz_engines.py 120 1400 --rules
run python tests/fuzz_ilqr_forms.py 40
