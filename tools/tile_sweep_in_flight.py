"""The traversal tile of k_backward3v was chosen with ONE plan on the chip (docs/NOTEBOOK.md 2.2b: 8x8 within 0.1 % of the best
of tile_v {2,4,8,16} x tile_a {4,8,16,32}).  With two plans in flight each launch holds half the slots: is 8x8 still the tile?
One child process per (tile_v, tile_a, skew) -- TPLHIP_DEBUG is read when a handle is initialised -- each timing one stream and
two streams on the headline grid.

    python tools/tile_sweep_in_flight.py [grid] [plans]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(grid, n):
    sys.path.insert(0, ROOT)
    import bench
    a = bench.dp_setup(grid, 0, 0)
    b = bench.dp_setup(grid, 0, 1)
    sweeps = a[2].t_steps - 1

    def run(planners):
        for _ in range(n // len(planners)):
            for q in planners:
                q[0].plan_resident(q[1])
        for q in planners:
            q[0].sync()

    out = {}
    for name, pl in (("one", (a,)), ("two", (a, b))):
        run(pl)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            run(pl)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        out[name] = round(best / (n * sweeps) * 1e6, 2)
    info = a[0].info()
    out["tile"] = [info.get("tile_v3", info.get("tile_v")), info.get("tile_a")]
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    grid = sys.argv[1] if len(sys.argv) > 1 else "target"
    n = sys.argv[2] if len(sys.argv) > 2 else "24"
    rows = []
    for skew in (0, 1):
        for tv in (2, 4, 8, 16):
            for ta in (4, 8, 16, 32):
                env = dict(os.environ, TPLHIP_DEBUG=f"tile_v={tv},tile_a={ta},skew={skew}")
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", grid, n], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                line = r.stdout.decode().strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "{}"
                rows.append(dict(json.loads(line), tile_v=tv, tile_a=ta, skew=skew))
                print(rows[-1], flush=True)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", grid, n], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    rows.append(dict(json.loads(r.stdout.decode().strip().splitlines()[-1]), default=True))
    print(rows[-1], flush=True)
    print(json.dumps({"grid": grid, "plans": int(n), "unit": "us per sweep", "rows": rows}))


if __name__ == "__main__":
    main()
