#!/usr/bin/env python3
"""A/B timing of the DP sweep under environment overrides, interleaved rounds in
ONE process (cdna_hip_programming.md rule 24).
usage: python3 tools/dp_ab.py grid "VAR=val" "VAR=val2" ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

grid = sys.argv[1]
variants = sys.argv[2:] or ["BASE=1"]
objs = []
for v in variants:
    for kv in v.split(","):
        k, val = kv.split("=", 1)
        os.environ[k] = val
    objs.append(bench.dp_setup(grid, 0))
    for kv in v.split(","):
        os.environ.pop(kv.split("=", 1)[0], None)
res = {v: [] for v in variants}
for rnd in range(5):
    for v, (dp, env, p, *_rest) in zip(variants, objs):
        dp.plan_resident(env)
        dp.sync()
        dp.timing_enable(True)
        for _ in range(3):
            dp.plan_resident(env)
        dp.sync()
        ms, n = dp.timing_get()
        dp.timing_enable(False)
        res[v].append(ms / n * 1e3)
for v in variants:
    r = sorted(res[v])
    print(objs[variants.index(v)][0].info())
    print(f"{v:40s} sweep us: min {r[0]:8.2f} median {r[len(r)//2]:8.2f} max {r[-1]:8.2f}")
