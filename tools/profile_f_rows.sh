#!/bin/bash
# Evidence for the SURVEY 8(f) kernels: kernel traces and one PMC pass each for the (s, ds, l) planner
# (k_ll_*), the lateral lattice (k_poly_lat_*) and the occupancy pipeline (k_raster_or, k_occ_to_frenet, ...); k_ll_backward's
# rate against the grid size; the same-box A/B of the slope-division pre-test.  On the GPU box, from the repo root:
#   bash tools/profile_f_rows.sh r05        -> gpurun_out/r05_f/...
set -u
TAG=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG}_f
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH="$REPO${PYTHONPATH:+:$PYTHONPATH}"
for D in lat_lon_plan poly_lat_only env_update_only; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$D" -- python3 "$REPO/tools/$D.py" > "$OUT/trace_$D.log" 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --output-format csv -d "$OUT/pmc_$D/pass1" -- python3 "$REPO/tools/$D.py" > "$OUT/pmc_$D.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS \
      --output-format csv -d "$OUT/pmc_$D/pass2" -- python3 "$REPO/tools/$D.py" >> "$OUT/pmc_$D.log" 2>&1
done
cd "$REPO"
for K in k_ll_backward k_ll_forward k_ll_prelude; do python3 tools/pmc_summary.py "gpurun_out/${TAG}_f/pmc_lat_lon_plan" $K all; done > "$OUT/pmc_lat_lon.txt" 2>&1
for K in k_poly_lat; do python3 tools/pmc_summary.py "gpurun_out/${TAG}_f/pmc_poly_lat_only" $K all; done > "$OUT/pmc_poly_lat.txt" 2>&1
for K in k_raster_or k_occ_to_frenet k_close k_dist; do python3 tools/pmc_summary.py "gpurun_out/${TAG}_f/pmc_env_update_only" $K all; done > "$OUT/pmc_env.txt" 2>&1
python3 tools/lat_lon_rate.py > "$OUT/lat_lon_rate.json" 2> "$OUT/lat_lon_rate.log"
# A/B: the slope division of evalAction behind a wave-uniform pre-test (shipped) against the division in every action
{
  for r in 1 2; do
    for so in "" tools/ab_tmp/tplhip_ll_nopretest.so; do
      printf "%-28s " "${so:-shipped (pre-test)}"
      TPLHIP_LIB="${so:+$REPO/$so}" python3 tools/lat_lon_plan.py
    done
  done
} > "$OUT/lat_lon_slope_ab.txt" 2>&1
(cd /tmp && TPLHIP_LIB="$REPO/tools/ab_tmp/tplhip_ll_nopretest.so" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_lat_lon_nopretest" -- python3 "$REPO/tools/lat_lon_plan.py" > "$OUT/trace_lat_lon_nopretest.log" 2>&1)
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
cat "$OUT/lat_lon_slope_ab.txt"; tail -n 12 "$OUT/lat_lon_rate.log"
