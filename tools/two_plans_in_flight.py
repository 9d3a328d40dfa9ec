"""Do the tails of consecutive sweeps leave room?  Two planners (own HIP stream each) plan the headline workload
(256x128x64, 50 stages) with their launches interleaved, against ONE planner doing the same number of plans back to back.
Sweeps of one plan depend on each other; sweeps of two plans do not, so the second plan's workgroups can take the slots
the first plan's sweep leaves free while its last workgroups drain.

    python tools/two_plans_in_flight.py [grid] [plans]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402


def main():
    grid = sys.argv[1] if len(sys.argv) > 1 else "target"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    a = bench.dp_setup(grid, 0, 0)
    b = bench.dp_setup(grid, 0, 1)
    c = bench.dp_setup(grid, 0, 2)
    sweeps = a[2].t_steps - 1

    def one(k):
        for _ in range(k):
            a[0].plan_resident(a[1])
        a[0].sync()

    def many(k, planners):
        for _ in range(k // len(planners)):
            for q in planners:
                q[0].plan_resident(q[1])
        for q in planners:
            q[0].sync()

    out = {"grid": grid, "plans": n}
    for name, fn in (("one_stream", lambda: one(n)), ("two_streams", lambda: many(n, (a, b))), ("three_streams", lambda: many(n, (a, b, c)))):
        fn()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        out[name] = {"ms": best * 1e3, "sweeps_per_s": n * sweeps / best, "us_per_sweep": best / (n * sweeps) * 1e6}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
