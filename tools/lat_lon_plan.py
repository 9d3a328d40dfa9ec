#!/usr/bin/env python3
"""Profiling driver: DynProgLatLonPlanner's plan on tpl's default grid (latency.lat_lon_update), 35 times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit import latency                         # noqa: E402
r = latency.lat_lon_update(0)
print(r["p50_ms"], r["p99_ms"], r["update_traj_dp_p50_ms"])
