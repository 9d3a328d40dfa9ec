#!/usr/bin/env python3
"""Profiling driver: tpl's default-grid plan through the drop-in surface (reinit + update), 60 times."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit import latency                         # noqa: E402
r = latency.dp_default_plan(0, plans=60)
print(r["p50_ms"], r["p99_ms"])
