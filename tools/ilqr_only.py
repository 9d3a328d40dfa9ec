#!/usr/bin/env python3
"""Profiling driver: the batched iLQR workload of bench.py and nothing else.  usage: python3 tools/ilqr_only.py [mpc|lateral] [batch] [path of an experimental build of the solver library]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit import ilqr as bench_ilqr  # noqa: E402

res = bench_ilqr.run(workload=sys.argv[1] if len(sys.argv) > 1 else "mpc", updates=3, warmup=1,
                     batch=int(sys.argv[2]) if len(sys.argv) > 2 else 1024, lib=sys.argv[3] if len(sys.argv) > 3 else None)
print("done", res["value"], res["kernel_ms_per_update"])
