#!/usr/bin/env python3
"""Profiling driver: the occupancy pipeline's per-cycle update (latency.env_update), 60 times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit import latency                         # noqa: E402
r = latency.env_update(0)
print(r["p50_ms"], r["p99_ms"], r["triangles"])
