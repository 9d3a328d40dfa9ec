"""Multi-GPU: independent plans shard across the GPUs of a node, one process per
GPU.  Nothing in the reference is distributed; this is the new
"8192 Monte-Carlo scenario rollouts sharded per GPU, all-gather of the best
plan" functionality of BASELINE configs[4].

The data path has no collective: every rank plans its contiguous block of
scenarios on its own GPU.  The only exchange is ONE all-gather of a fixed-size
best-plan record per rank (a few KB: latency-bound over xGMI, so it is skipped
entirely when world_size == 1), followed by the same lexicographic argmin on
every rank.  Backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
"""

import numpy as np


def shard_range(n, rank, world):
    """Contiguous block partition of n units: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def plan_record(traj_np, scenario_id):
    """Fixed-size record of one plan: [constr, cost, id, t0,s0,v0,a0,j0,cost0,constr0, ...].

    traj_np: float32[t_steps, 7] = LonState rows.  `constr` is the validity
    measure of the Python caller, max over states[1:-1]
    (planning/dyn_prog/poly_lat_dp_lon_planner.py:99-101); `cost` is the cost of
    the first node, i.e. the value of the whole plan."""
    traj_np = np.asarray(traj_np, dtype=np.float32)
    inner = traj_np[1:-1, 6] if traj_np.shape[0] > 2 else traj_np[:, 6]
    head = np.array([float(np.max(inner)) if inner.size else 0.0, float(traj_np[0, 5]), float(scenario_id)],
                    dtype=np.float64)
    return np.concatenate([head, traj_np.astype(np.float64).ravel()])


def record_from_device(rec):
    """The device-side record of ``DynProgLonPlanner.best_record()`` -- (constr, cost, id, float32[t_steps, 7]) --
    in the layout of ``plan_record``."""
    constr, cost, rid, states = rec
    return np.concatenate([np.array([constr, cost, float(rid)], dtype=np.float64), np.asarray(states, dtype=np.float64).ravel()])


def better(a, b):
    """Lexicographic order of the planner itself (lon_planner.cu:153-170):
    smaller constraint violation first, then smaller cost, then smaller id."""
    return (a[0], a[1], a[2]) < (b[0], b[1], b[2])


def local_best(records):
    best = None
    for r in records:
        if best is None or better(r, best):
            best = r
    return best


def allgather_best(record, dist=None, device=None):
    """All ranks contribute their best record; every rank returns the global
    best (identical on all ranks).  dist: torch.distributed or None (single
    process: no collective at all)."""
    if dist is None or not dist.is_initialized():
        return record
    import torch
    world = dist.get_world_size()
    t = torch.as_tensor(np.asarray(record, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous())
    return local_best([o.cpu().numpy() for o in out])
