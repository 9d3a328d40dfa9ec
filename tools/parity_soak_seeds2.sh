run() { echo "  $*"; ( "$@" 2>&1 | tail -2 | sed 's/^/        /' ); }
echo "second soak, fresh seeds"
run python tests/fuzz_dp.py 1500 170000
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 600 171000
run python tests/fuzz_dp.py 500 172000 --rules
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 250 173000 --rules
TPLHIP_DEBUG=skew=1,big=1 run python tests/fuzz_dp.py 300 174000
run python tests/fuzz_scenarios.py 100 1500
run python tests/fuzz_engines.py 80 1200
run python tests/fuzz_ilqr_forms.py 40
