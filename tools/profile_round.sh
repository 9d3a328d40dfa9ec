#!/bin/bash
# End-of-round evidence run (on the GPU box, from the repo root):
#   bash tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...
set -u
TAG=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
python bench.py --steps 20 --warmup 3 > "$OUT/bench_target.json" 2> "$OUT/bench_target.err"
python bench.py --steps 20 --warmup 3 --grid config2 --no-cpu-baseline --no-secondary > "$OUT/bench_config2.json" 2>/dev/null
python bench.py --steps 5 --warmup 1 --grid config4 --no-cpu-baseline --no-secondary > "$OUT/bench_config4.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH="$REPO${PYTHONPATH:+:$PYTHONPATH}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dp" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/trace_dp.log" 2>&1
# (the default line keeps two plans in flight: the sweep's duration in that trace is its duration WHILE sharing the chip; the same
# command with one plan in flight gives the kernel alone -- `roofline.alone` of the line)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dp_alone" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --in-flight 1 > "$OUT/trace_dp_alone.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr" -- python3 "$REPO/tools/ilqr_only.py" mpc > "$OUT/trace_ilqr.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_mc" -- python3 "$REPO/tools/mc_only.py" > "$OUT/trace_mc.log" 2>&1
cd "$REPO"
bash tools/pmc_dp.sh "gpurun_out/$TAG/pmc_target" target > "$OUT/pmc_target.txt" 2>&1
bash tools/pmc_dp.sh "gpurun_out/$TAG/pmc_config4" config4 > "$OUT/pmc_config4.txt" 2>&1
bash tools/pmc_dp.sh "gpurun_out/$TAG/pmc_mc" mc > "$OUT/pmc_mc.txt" 2>&1
bash tools/pmc_ilqr.sh "gpurun_out/$TAG/pmc_ilqr" mpc > "$OUT/pmc_ilqr.txt" 2>&1
# the two-wavefront forms of the batched iLQR update (4096 problems): kernel traces (fused: one launch; split: with the gaps
# between its launches), PMC of the fused kernel and of the split form's chain and refresh kernels; and the same-box A/B of
# all three forms over the batch size
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr_fused" -- python3 "$REPO/tools/ilqr_only.py" mpc 4096 > "$OUT/trace_ilqr_fused.log" 2>&1)
bash tools/pmc_ilqr.sh "gpurun_out/$TAG/pmc_ilqr_fused" mpc 4096 k_ilqr_fused > "$OUT/pmc_ilqr_fused.txt" 2>&1
(cd /tmp && TPLHIP_DEBUG=ilqr_split=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ilqr_split" -- python3 "$REPO/tools/ilqr_only.py" mpc 4096 > "$OUT/trace_ilqr_split.log" 2>&1)
python3 tools/trace_gaps.py "$OUT/trace_ilqr_split" k_split > "$OUT/ilqr_split_gaps_4096.txt"
TPLHIP_DEBUG=ilqr_split=1 bash tools/pmc_ilqr.sh "gpurun_out/$TAG/pmc_ilqr_split" mpc 4096 k_split_chain > "$OUT/pmc_ilqr_split_chain.txt" 2>&1
python3 tools/pmc_summary.py "gpurun_out/$TAG/pmc_ilqr_split" k_split_refresh all > "$OUT/pmc_ilqr_split_refresh.txt" 2>&1
python3 tools/ab_ilqr_split.py mpc --forms 0,1,2 1024 1536 2048 4096 8192 16384 > "$OUT/ilqr_split_ab.txt" 2>&1
python3 -c "import json, bench; json.dump(bench.dp_general_sweep_price(0), open('$OUT/dp_general_sweep.json', 'w'), indent=1)" > /dev/null 2>&1
# the SURVEY 8(f) kernels: traces, one PMC pass each, k_ll_backward against the grid size (-> gpurun_out/${TAG}_f)
bash tools/profile_f_rows.sh "$TAG" > "$OUT/f_rows.log" 2>&1
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
tail -n 3 "$OUT"/bench_*.json | cut -c1-600
cat "$OUT/pmc_target.txt" | tail -32
