#!/bin/bash
# Pricing the two re-mappings of the iLQR chain the round-4 review asked for, same box, one process per line:
#   (a)  two problems per wavefront in the line-search rollouts;
#   (b1) the stage-cost pass folded into the rollout quads;  (b2) k_ilqr_update rebuilt as per-phase calls at ONE wavefront per SIMD.
# The prices come from builds that do a part TWICE (TPL_ILQR_ABLATE = 5: the first half of the stages rolled out again, 6: every
# candidate stage cost evaluated twice): results and decisions are unchanged (the same bits), the added time is the time of the part.
# Builds that LEAVE a part out (3: no candidate stage costs, 4: rollouts over half the stages) change which steps are accepted and
# how often the derivatives are refreshed: listed as bounds only.
# Here:  for v in 3 4 5 6; do python tools/ab_ilqr_split.py build abl$v -DTPL_ILQR_ABLATE=$v; done
# Box:   bash tools/ab_ilqr_remap.sh > gpurun_out/ilqr_remap_ab.txt
set -u
echo "## 1 024 problems (BASELINE configs[2]), one wavefront per SIMD: k_ilqr_update (single) and (b2) the phases as calls at one"
echo "## wavefront per SIMD (fused1 = TPLHIP_DEBUG=ilqr_split=3); abl5 - shipped = half a rollout, abl6 - shipped = one cost pass"
python tools/ab_ilqr_split.py mpc --libs shipped,abl5,abl6,abl3,abl4 --forms 0 1024
python tools/ab_ilqr_split.py mpc --libs shipped --forms 3 1024
python tools/ab_ilqr_split.py mpc --libs shipped --forms 0,3 512
echo "## 4 096 problems, the fused form (two wavefronts per SIMD)"
python tools/ab_ilqr_split.py mpc --libs shipped,abl5,abl6,abl3,abl4 --forms 2 4096
echo "## ... with the trajectory mirrors OUT of LDS (10 KB per problem: Lds block + reference arrays) -- what four problems in flight"
echo "##     per SIMD (two wavefronts x two problems each) would have to run on: 2 x 20 KB per wavefront leave one wavefront per SIMD"
TPLHIP_DEBUG=ilqr_split_lds_kb=10 python tools/ab_ilqr_split.py mpc --libs shipped,abl5 --forms 2 4096
echo "## repeat of the first line (box drift)"
python tools/ab_ilqr_split.py mpc --libs shipped --forms 0 1024
