#!/bin/bash
# Randomised parity soaks of a round's last revision (GPU box, repo root):  bash tools/parity_soak.sh > gpurun_out/<tag>/parity_soak.txt
run() { echo "  $*"; ( "$@" 2>&1 | tail -2 | sed 's/^/        /' ); }
echo "DP (s,v,a) planner, HIP vs oracle/dp_lon_oracle.c, tables of every stage + 3 trajectories per configuration, bit for bit"
run python tests/fuzz_dp.py 1000 60000
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 500 61000
echo "  (the next four: TPLHIP_DEBUG = no_v3,no_s3 | no_fastdiv | div_form=5 | skew=1,big=1)"
TPLHIP_DEBUG=no_v3,no_s3 run python tests/fuzz_dp.py 100 62000
TPLHIP_DEBUG=no_fastdiv run python tests/fuzz_dp.py 200 63000
TPLHIP_DEBUG=div_form=5 run python tests/fuzz_dp.py 200 64000
TPLHIP_DEBUG=skew=1,big=1 run python tests/fuzz_dp.py 300 65000
echo "  (random arithmetic rules: R2 x R3 x R1, each case against the oracle build of its rule)"
run python tests/fuzz_dp.py 300 66000 --rules
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 150 67000 --rules
echo "  (... and R3's blend order / eight-corner sums on top: 72 combinations; k_backward3o on the large-grid legs)"
run python tests/fuzz_dp.py 400 68000 --rules --orders
TPLHIP_DEBUG=big=1 run python tests/fuzz_dp.py 300 69000 --rules --orders
TPLHIP_DEBUG=skew=1,big=1 run python tests/fuzz_dp.py 150 69500 --rules --orders
echo "Scenario batches + device-side best-plan record"
run python tests/fuzz_scenarios.py 60 500
echo "Occupancy pipeline (f-1), lateral lattice (f-2), (s,ds,l) planner (f-3)"
run python tests/fuzz_engines.py 40 200
echo "  (the (s,ds,l) planner under random texture-filter rules: weight rule x blend form x blend order)"
run python tests/fuzz_engines.py 60 300 --rules
echo "iLQR, HIP vs oracle/ilqr_oracle.cpp, 7 problems x 10 seeds; then the same through the split-launch and the fused form"
runl() { echo "  $*"; ( "$@" 2>&1 | tail -12 | sed 's/^/        /' ); }
runl python tests/fuzz_ilqr.py 10
echo "  (TPLHIP_DEBUG=ilqr_split=1)"
TPLHIP_DEBUG=ilqr_split=1 runl python tests/fuzz_ilqr.py 10
echo "  (TPLHIP_DEBUG=ilqr_split=2)"
TPLHIP_DEBUG=ilqr_split=2 runl python tests/fuzz_ilqr.py 10
echo "iLQR launch forms against each other (single / split / fused), ragged batches, random integrator and order, bit for bit"
run python tests/fuzz_ilqr_forms.py 25
echo "  (the doubled-part builds of tools/ab_ilqr_remap.sh give the shipped bits: TPL_ILQR_ABLATE = 5, 6 against the shipped library, one ragged batch each)"
python - <<'PY'
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import fuzz_ilqr_forms as F
from tpl_amd.optim import genopt, solver
ref, _ = F.run("trajectory_tracking_mpc", "mpc_straight_fixed_iters", 7, 0)
orig = genopt.build
for tag in ("abl5", "abl6"):
    so = os.path.abspath(f"tools/ab_tmp/ilqr_{tag}.so")
    if not os.path.exists(so):
        print(f"        {tag}: not built"); continue
    genopt.build = lambda cfg, name=None, out_dir=None, so=so: solver.factory(so, cfg.param_defaults)
    for form in (0, 2):
        got, _ = F.run("trajectory_tracking_mpc", "mpc_straight_fixed_iters", 7, form)
        bad = [f for a, b in zip(ref, got) for f in F.FIELDS if not np.array_equal(a[f], b[f], equal_nan=True)]
        print(f"        {tag}, form {form}: {len(bad)} field mismatches against the shipped single launch")
genopt.build = orig
PY
