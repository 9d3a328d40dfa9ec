#!/usr/bin/env python3
"""Kernel timeline of the LAST default-grid plan in a rocprofv3 --kernel-trace run of tools/dp_default_plan.py:
   (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python3 $R/tools/dp_default_plan.py)
   python3 tools/plan_timeline.py gpurun_out/tl
start and duration in us from the first kernel of the plan (memory copies are not in a kernel trace)."""
import csv
import glob
import re
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_forward" in r["Kernel_Name"]]
last, prev = ends[-1], ends[-2]
plan = rows[prev + 1:last + 1]
t0 = int(plan[0]["Start_Timestamp"])
busy = 0
for r in plan:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
    print(f"  {(name.group(1) if name else r['Kernel_Name'][:44]):46s} start {(s - t0) / 1e3:7.1f}  {(e - s) / 1e3:6.1f}")
print(f"  -> {busy / 1e3:.0f} us of kernels in a span of {(int(plan[-1]['End_Timestamp']) - t0) / 1e3:.0f} us")
