#!/bin/bash
# End-of-round evidence run (on the GPU box, from the repo root):
#   bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 3 > "$OUT/bench_target.json" 2> "$OUT/bench_target.err"
python bench.py --steps 20 --warmup 3 --grid config2 --no-cpu-baseline --no-secondary > "$OUT/bench_config2.json" 2>/dev/null
python bench.py --steps 5 --warmup 1 --grid config4 --no-cpu-baseline --no-secondary > "$OUT/bench_config4.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH="$REPO${PYTHONPATH:+:$PYTHONPATH}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dp" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/trace_dp.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr" -- python3 "$REPO/tools/ilqr_only.py" mpc > "$OUT/trace_ilqr.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mc" -- python3 "$REPO/tools/mc_only.py" > "$OUT/trace_mc.log" 2>&1
cd "$REPO"
bash tools/pmc_dp.sh "gpurun_out/$TAG/pmc_target" target > "$OUT/pmc_target.txt" 2>&1
bash tools/pmc_dp.sh "gpurun_out/$TAG/pmc_config4" config4 > "$OUT/pmc_config4.txt" 2>&1
bash tools/pmc_dp.sh "gpurun_out/$TAG/pmc_mc" mc > "$OUT/pmc_mc.txt" 2>&1
bash tools/pmc_ilqr.sh "gpurun_out/$TAG/pmc_ilqr" mpc > "$OUT/pmc_ilqr.txt" 2>&1
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
tail -n 3 "$OUT"/bench_*.json | cut -c1-600
cat "$OUT/pmc_target.txt" | tail -32
