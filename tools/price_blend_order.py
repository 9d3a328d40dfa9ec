#!/usr/bin/env python3
"""us per stage sweep under every arithmetic rule the planner can follow (bench.py's secondary.general_sweep leg) on the
three grids of DESIGN.md 0: the headline 256x128x64, tpl's default 201x37x20 and configs[1] 128x64x32.

    python tools/price_blend_order.py > gpurun_out/price_blend_order.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out = {}
for grid in ("target", "default", "config2"):
    r = bench.dp_general_sweep_price(0, grid=grid, plans=5)
    r.pop("selection", None)
    out[grid] = r
    print(grid, {k: (v["kernel"], round(v["us_per_sweep"], 2)) for k, v in r.items() if isinstance(v, dict)}, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
