"""N > 1 path on CPU: world_size-2 gloo processes shard a set of scenarios, each
rank reduces its block to a best-plan record, one all-gather, identical global
argmin on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_plan(sid, t_steps=6):
    """Stand-in for one planner result (the HIP planner needs a GPU): a
    deterministic float32[t_steps, 7] trajectory whose cost/constr depend on the id."""
    rng = np.random.default_rng(1000 + sid)
    tr = rng.uniform(0, 1, size=(t_steps, 7)).astype(np.float32)
    tr[:, 6] = 0.0 if sid % 3 else 0.5          # every third scenario violates a constraint
    tr[0, 5] = 50.0 + ((sid * 37) % 17)          # plan cost
    return tr


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    from tpl_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(n, rank, world)
    best = sharding.local_best([sharding.plan_record(_fake_plan(i), i) for i in range(lo, hi)])
    glob = sharding.allgather_best(best, dist)
    q.put((rank, lo, hi, glob))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_everything():
    from tpl_amd import sharding
    for n in (0, 1, 7, 8, 8192, 1000):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_allgather_of_best_plan():
    sys.path.insert(0, ROOT)
    from tpl_amd import sharding
    n, world = 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    serial = sharding.local_best([sharding.plan_record(_fake_plan(i), i) for i in range(n)])
    for rank, lo, hi, glob in results:
        assert np.array_equal(glob, serial), rank
    assert serial[0] == 0.0                      # a constraint-free plan wins over any cheaper violating one
    spans = sorted((lo, hi) for _, lo, hi, _ in results)
    assert spans[0][0] == 0 and spans[-1][1] == n and spans[0][1] == spans[1][0]


def test_single_process_skips_the_collective():
    from tpl_amd import sharding
    rec = sharding.plan_record(_fake_plan(3), 3)
    assert sharding.allgather_best(rec, None) is rec
