#!/usr/bin/env python3
"""Where a planner plugin's cycle goes on the host: cProfile over 200 cycles of PolyLatDpLonPlanner.update /
DpLatLonPlanner.update on the drive of benchkit/drive.py (engines on the GPU).  usage: prof_planner_cycle.py [poly|latlon]"""
import cProfile
import os
import pstats
import sys
from contextlib import contextmanager

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchkit import drive                                   # noqa: E402
from tpl_amd.planning.dyn_prog import DpLatLonPlanner, PolyLatDpLonPlanner   # noqa: E402
from tpl_amd.store import Bundle                            # noqa: E402


@contextmanager
def lock():
    yield


cls = DpLatLonPlanner if (len(sys.argv) > 1 and sys.argv[1] == "latlon") else PolyLatDpLonPlanner
pl = cls(Bundle(), lock)
d = drive.Drive(v_ego=9.0)
cycles = [d.cycle(k) for k in range(220)]
for sh in cycles[:20]:
    pl.update(sh)
pr = cProfile.Profile()
pr.enable()
for sh in cycles[20:]:
    pl.update(sh)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
