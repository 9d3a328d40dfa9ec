set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05d; mkdir -p "$OUT"
python bench.py --steps 20 --warmup 3 > "$OUT/bench_target.json" 2> "$OUT/bench_target.err"
python bench.py --steps 20 --warmup 3 --in-flight 1 --no-cpu-baseline --no-secondary > "$OUT/bench_target_one_in_flight.json" 2>/dev/null
python bench.py --steps 20 --warmup 3 --in-flight 1 --chained 0 --no-cpu-baseline --no-secondary > "$OUT/bench_target_one_in_flight_stage_by_stage.json" 2>/dev/null
python bench.py --steps 20 --warmup 3 --grid config2 --no-cpu-baseline --no-secondary > "$OUT/bench_config2.json" 2>/dev/null
python bench.py --steps 5 --warmup 1 --grid config4 --no-cpu-baseline --no-secondary > "$OUT/bench_config4.json" 2>/dev/null
python bench.py --steps 5 --warmup 1 --grid config4 --in-flight 1 --no-cpu-baseline --no-secondary > "$OUT/bench_config4_one_in_flight.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH="$REPO${PYTHONPATH:+:$PYTHONPATH}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dp" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-form-ab > "$OUT/trace_dp.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dp_alone" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --in-flight 1 --chained 0 --no-form-ab > "$OUT/trace_dp_alone.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_dp_chained" -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --in-flight 1 --no-form-ab > "$OUT/trace_dp_chained.log" 2>&1
cd "$REPO"
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
for f in "$OUT"/bench_*.json; do echo $f; cut -c1-400 $f; done
find "$OUT" -name "*kernel_stats.csv" | while read f; do echo $f; head -4 "$f"; done
