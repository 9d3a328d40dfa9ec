#!/usr/bin/env python3
"""Time the DP sweep of whatever library TPLHIP_LIB points at.  usage: dp_time.py [grid]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
grid = sys.argv[1] if len(sys.argv) > 1 else "target"
dp, env, p, *_ = bench.dp_setup(grid, 0)
if os.environ.get("TPL_AB_R1") == "0":
    dp.set_contraction(False)
res = []
for _ in range(5):
    dp.plan_resident(env); dp.sync()
    dp.timing_enable(True)
    for _ in range(3):
        dp.plan_resident(env)
    dp.sync()
    ms, n = dp.timing_get(); dp.timing_enable(False)
    res.append(ms / n * 1e3)
res.sort()
print(os.path.basename(os.environ.get("TPLHIP_LIB", "default")), dp.info(), f"sweep us min {res[0]:.2f} median {res[2]:.2f}")
