#!/usr/bin/env python3
"""Same-box, same-process A/B of the sweep under the two settings of rule R1 (and any other rule): planners that differ
only in their arithmetic rule, timed in interleaved rounds (cdna_hip_programming.md rule 24), alone on the chip and with
two planners of the SAME rule in flight.
usage: python3 tools/ab_rules.py [grid] [rounds] > profiles/r06_dp_contraction_ab.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

grid = sys.argv[1] if len(sys.argv) > 1 else "target"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
RULES = {"R1=1 (default, contracted)": dict(fma=1), "R1=0 (one rounding per operation)": dict(fma=0)}


def make(rule):
    dp, env, p, *_ = bench.dp_setup(grid, 0)
    dp.set_contraction(bool(rule["fma"]))
    dp.plan_resident(env)
    dp.sync()
    return dp, env, p


fleet = {name: [make(r), make(r)] for name, r in RULES.items()}
alone = {n: [] for n in RULES}
pair = {n: [] for n in RULES}
for rnd in range(rounds):
    for name, ((dp, env, p), (dp2, env2, _)) in fleet.items():
        dp.timing_enable(True)
        for _ in range(3):
            dp.plan_resident(env)
        dp.sync()
        ms, n = dp.timing_get()
        dp.timing_enable(False)
        alone[name].append(ms / n * 1e3)
        t0 = time.perf_counter()
        k = 8
        for _ in range(k):
            dp.plan_resident(env)
            dp2.plan_resident(env2)
        dp.sync()
        dp2.sync()
        pair[name].append((time.perf_counter() - t0) / (2 * k * (p.t_steps - 1)) * 1e6)
out = {"grid": grid, "rounds": rounds, "what": "us per stage sweep; alone = HIP events around one plan's sweeps on one stream, "
       "two_in_flight = wall time of 16 plans on two planners / their sweeps (fronts and rollouts included)"}
for name in RULES:
    a, b = sorted(alone[name]), sorted(pair[name])
    info = fleet[name][0][0].info()
    out[name] = {"kernel": info["sweep"], "division_forms": [info["path_div3"], info["range_div3"]], "rule": info["rule"],
                 "alone_us": {"min": a[0], "median": a[len(a) // 2], "max": a[-1], "all": alone[name]},
                 "two_in_flight_us": {"min": b[0], "median": b[len(b) // 2], "max": b[-1], "all": pair[name]}}
print(json.dumps(out, indent=1))
