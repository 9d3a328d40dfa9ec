#!/bin/bash
# Round 6, last evidence pass on the final libraries (on the GPU box, from the repo root): the bench lines, the solver's kernel trace / PMC
# after the ring change, its phase stamps at 1 024 problems and for one MPC tick.   bash tools/round6_final_evidence.sh
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06f; mkdir -p "$OUT"
python bench.py --steps 20 --warmup 3 > "$OUT/bench_target.json" 2> "$OUT/bench_target.err"
python bench.py --steps 20 --warmup 3 --in-flight 1 --no-cpu-baseline --no-secondary > "$OUT/bench_target_one_in_flight.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH="$REPO${PYTHONPATH:+:$PYTHONPATH}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr" -- python3 "$REPO/tools/ilqr_only.py" mpc > "$OUT/trace_ilqr.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr_fused" -- python3 "$REPO/tools/ilqr_only.py" mpc 4096 > "$OUT/trace_ilqr_fused.log" 2>&1
cd "$REPO"
bash tools/pmc_ilqr.sh "gpurun_out/r06f/pmc_ilqr" mpc > "$OUT/pmc_ilqr.txt" 2>&1
bash tools/pmc_ilqr.sh "gpurun_out/r06f/pmc_ilqr_fused" mpc 4096 k_ilqr_fused > "$OUT/pmc_ilqr_fused.txt" 2>&1
python3 tools/ab_ilqr_split.py mpc --forms 0,1,2 1024 2048 4096 8192 16384 > "$OUT/ilqr_forms_ab.txt" 2>&1
bash tools/ilqr_stamps.sh 1024 > "$OUT/ilqr_stamps_1024.txt" 2>&1
python tools/ilqr_stamp_run.py 1 20 300 > "$OUT/ilqr_stamps_mpc_tick.txt" 2>&1
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
cut -c1-300 "$OUT/bench_target.json"; echo; cut -c1-200 "$OUT/bench_target_one_in_flight.json"; echo; tail -3 "$OUT/ilqr_stamps_mpc_tick.txt"
