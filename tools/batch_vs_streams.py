"""Why do two plans in flight beat one (tools/two_plans_in_flight.py: 40.6 -> 35.4 us per 256x128x64 sweep)?
If it is the drain of a sweep's second residency round that the other plan fills, then TWO SCENARIOS PER LAUNCH on ONE
stream (blockIdx.y = scenario: one launch of 8 192 workgroups, four rounds, one drain per two sweeps) must gain about as
much; if it gains nothing, the two streams' gain is something a single launch does not have (two dispatch queues,
workgroups of two launches out of step on a CU).  Same box, same process:

    python tools/batch_vs_streams.py [grid] [plans]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402


def main():
    import tpl_amd
    from benchkit import scenarios as sc
    grid = sys.argv[1] if len(sys.argv) > 1 else "target"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    a = bench.dp_setup(grid, 0, 0)
    b = bench.dp_setup(grid, 0, 1)
    p, ep, path = a[2], a[3], a[4]
    sweeps = p.t_steps - 1
    out = {"grid": grid, "plans": n, "sweep": a[0].info()["sweep"]}

    def timed(fn):
        fn()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        return {"ms": best * 1e3, "us_per_sweep": best / (n * sweeps) * 1e6, "sweeps_per_s": n * sweeps / best}

    def one():
        for _ in range(n):
            a[0].plan_resident(a[1])
        a[0].sync()

    def two():
        for _ in range(n // 2):
            a[0].plan_resident(a[1])
            b[0].plan_resident(b[1])
        a[0].sync()
        b[0].sync()

    out["one_stream"] = timed(one)
    out["two_streams"] = timed(two)
    del b
    for chunk in (2, 4):
        dp = tpl_amd.DynProgLonPlanner(0)
        dp.reinit_buffers(p)
        occs, inits = bench.mc_scenarios(ep, 0, chunk)
        dp.upload_scenarios(occs, ep, inits, path, chunk=chunk)

        def batched():
            for _ in range(n // chunk):
                dp.plan_scenarios(0, chunk, id_base=0, reset_best=True)
            dp.sync()

        out[f"one_stream_{chunk}_scenarios_per_launch"] = dict(timed(batched), sweep=dp.info()["sweep"])
        del dp
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
