#!/usr/bin/env python3
"""Factor between the bytes tools/micro/fetch_cal.hip moved (1 GiB per kernel) and what rocprofv3 reports.
usage: python3 tools/fetch_cal_summary.py <rocprofv3 output dir(s)...>"""
import csv
import glob
import os
import sys
from collections import defaultdict

MOVED = float(1 << 30)
acc = defaultdict(list)
for root in sys.argv[1:]:
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "k_read" not in name and "k_write" not in name:
                continue
            short = name[name.index("k_"):].split("(")[0]
            acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("# kernel, counter, value per dispatch, bytes moved / (value x unit)")
for (k, c), vals in sorted(acc.items()):
    for v in vals:
        unit = 1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else 64.0
        print(f"{k:16s} {c:22s} {v:16.1f}   moved/(value*{int(unit)}) = {MOVED / max(v * unit, 1.0):.4f}")
