#!/usr/bin/env python3
"""Per-kernel duration and the idle gap in front of each launch, from a rocprofv3 --kernel-trace CSV.
usage: python3 tools/trace_gaps.py <dir or *_kernel_trace.csv> [name filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for s, e, n in rows:
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if flt and flt not in short:
        prev_end = e
        continue
    dur[short] += e - s
    if prev_end is not None:
        gap[short] += max(0, s - prev_end)
    cnt[short] += 1
    prev_end = e
print(f"{'kernel':60s} {'calls':>7s} {'avg us':>10s} {'avg gap before us':>18s} {'total ms':>10s}")
for n in sorted(dur, key=lambda k: -dur[k]):
    print(f"{n[:60]:60s} {cnt[n]:7d} {dur[n] / cnt[n] / 1e3:10.2f} {gap[n] / cnt[n] / 1e3:18.2f} {dur[n] / 1e6:10.3f}")
