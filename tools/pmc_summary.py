#!/usr/bin/env python3
"""Average every PMC counter over the dispatches of kernels whose name contains
<substr>.  usage: python3 tools/pmc_summary.py <dir with pass*/> <substr> [all]
("all": keep kernels instantiated with a leading `true` template argument, which the DP summaries skip)"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, sub = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r.get("Kernel_Name", ""):
            continue
        if "<true" in r["Kernel_Name"] and "all" not in sys.argv[3:]:          # skip the trivial terminal-stage variant
            continue
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
print(f"# per-dispatch averages for kernels matching {sub!r}")
for k in sorted(acc):
    s, n = acc[k]
    print(f"{k:34s} {s / max(n, 1):18.1f}   (n={n})")
