#!/usr/bin/env python3
"""Wall-clock shares of a planner plugin's cycle (perf_counter around the named calls, no profiler):
usage: time_planner_parts.py [poly|latlon]"""
import os
import sys
import time
from collections import defaultdict
from contextlib import contextmanager

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tpl_amd                                              # noqa: E402
from benchkit import drive                                   # noqa: E402
from tpl_amd.planning import _kit                           # noqa: E402
from tpl_amd.planning.dyn_prog import DpLatLonPlanner, PolyLatDpLonPlanner, env_feed, plugins   # noqa: E402
from tpl_amd.store import Bundle                            # noqa: E402

acc = defaultdict(list)


def timed(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or f"{getattr(owner, '__name__', owner)}.{name}"

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label].append((time.perf_counter() - t0) * 1e3)
    setattr(owner, name, wrapper)


E = tpl_amd.DynProgEnvironment
for n in ("update", "get_occ_map_cartesian", "get_occ_map_slices", "get_dist_map_lon_slices", "set_ref_line", "insert_geometry", "reinit_buffers"):
    timed(E, n, "env." + n)
for n in ("update", "reinit_buffers", "reeval_traj"):
    timed(tpl_amd.DynProgLonPlanner, n, "lon." + n)
for n in ("update", "reinit_buffers"):
    timed(tpl_amd.PolyLatPlanner, n, "polylat." + n)
timed(env_feed.DpEnv, "update", "DpEnv.update")
timed(env_feed.DpEnv, "write_debug_data", "DpEnv.write_debug_data")
timed(env_feed.DpEnv, "update_environment", "DpEnv.update_environment")
timed(env_feed.DpEnv, "update_reference_line", "DpEnv.update_reference_line")
timed(_kit.EnvSnapshot, "take", "EnvSnapshot.take")
timed(env_feed, "gen_prediction_geometry", "gen_prediction_geometry")
for cls in (plugins.PolyLatDpLonPlanner, plugins.DpLatLonPlanner):
    for n in ("update_params", "check_replan", "shift_trajectory", "_plan", "update_trajectory", "write_debug_data"):
        if hasattr(cls, n):
            timed(cls, n, f"{cls.__name__}.{n}")


if os.environ.get("NO_ENV_DEBUG"):
    env_feed.DpEnv.write_debug_data = lambda self: None
if os.environ.get("NO_CART"):
    E.get_occ_map_cartesian = lambda self: None


@contextmanager
def lock():
    yield


cls = DpLatLonPlanner if (len(sys.argv) > 1 and sys.argv[1] == "latlon") else PolyLatDpLonPlanner
pl = cls(Bundle(), lock)
d = drive.Drive(v_ego=9.0)
cycles = [d.cycle(k) for k in range(220)]
for sh in cycles[:20]:
    pl.update(sh)
acc.clear()
tot = []
for sh in cycles[20:]:
    t0 = time.perf_counter()
    pl.update(sh)
    tot.append((time.perf_counter() - t0) * 1e3)
n = len(tot)
print(f"cycle: mean {np.mean(tot):.3f} ms, p50 {np.percentile(tot, 50):.3f} ms over {n} cycles")
slow = {k: [i for i, x in enumerate(v) if x > 5.0] for k, v in acc.items()}
print("calls over 5 ms:", {k: v for k, v in slow.items() if v})
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:44s} {sum(v) / n:7.3f} ms/cycle   {len(v) / n:5.2f} calls/cycle   p50 {np.percentile(v, 50) * 1e3:8.1f} us   max {np.max(v) * 1e3:9.1f} us (call {int(np.argmax(v))})")
