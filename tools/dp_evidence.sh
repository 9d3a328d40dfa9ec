#!/bin/bash
# Evidence for the DP sweep after a kernel change (GPU box, repo root): bench lines + kernel traces (tools/headline_evidence.sh,
# -> gpurun_out/r05d) and the PMC passes of the sweep on the three workloads (-> gpurun_out/r05d/pmc_*).
set -u
bash tools/headline_evidence.sh > gpurun_out/headline_evidence.log 2>&1
for w in target config4 mc; do bash tools/pmc_dp.sh gpurun_out/r05d/pmc_$w $w > gpurun_out/r05d/pmc_$w.txt 2>&1; done
find gpurun_out/r05d -name "*_agent_info.csv" -delete
find gpurun_out/r05d -name "*counter_collection.csv" -size +2M -delete
tail -5 gpurun_out/headline_evidence.log | cut -c1-300
tail -30 gpurun_out/r05d/pmc_target.txt
