"""N > 1 path on CPU: world_size-2 gloo processes shard a set of scenarios, each rank plans its block (here with the
CPU oracle -- the HIP planner needs a GPU; tests/test_dp_scenarios_gpu.py checks that its device-side record equals
this host-side reduction), reduces it to a best-plan record, one all-gather, identical global argmin on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GRID = (33, 9, 7, 5)          # s, v, a, stages: an oracle plan takes milliseconds


def _scenario(sid):
    """Scenario `sid` of a small Monte-Carlo set: own obstacles and initial state, seeded by the id."""
    from benchkit import scenarios as sc
    S, V, A, T = GRID
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=101, l_steps=21)
    rng = np.random.default_rng(4000 + sid)
    obst = () if sid % 5 == 0 else sc.random_obstacles(seed=sid, k=2 + sid % 3)
    init = sc.initial_state(0.0, float(rng.uniform(2.0, 14.0)), float(rng.uniform(-1.0, 1.0)))
    return p, ep, sc.straight_path(p.path_steps, p.path_step_size), sc.occ_map(ep, obst), init


def _oracle_record(sid):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_dp as O
    from tpl_amd import sharding
    p, ep, path, occ, init = _scenario(sid)
    _, trajs, _ = O.update(O.cparams(p), O.cparams(ep), path, occ, [init], nthreads=1, keep_tables=False)
    return sharding.plan_record(trajs[0], sid)


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    from tpl_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(n, rank, world)
    best = sharding.local_best([_oracle_record(i) for i in range(lo, hi)])
    glob = sharding.allgather_best(best, dist)
    q.put((rank, lo, hi, glob))
    dist.barrier()
    dist.destroy_process_group()


def _worker8(rank, world, port, n, q):
    """One rank of the 8-rank rehearsal: its block of scenario ids, the local reduction, ONE all-gather, and what the bench's
    workers assert before timing -- the process group really holds `world` ranks (an all-reduce of ones)."""
    sys.path.insert(0, ROOT)
    from tpl_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    lo, hi = sharding.shard_range(n, rank, world)
    best = sharding.local_best([_oracle_record(i) for i in range(lo, hi)])
    glob = sharding.allgather_best(best, dist)
    q.put((rank, lo, hi, int(ones.item()), best, glob))
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
    assert [sharding.shard_range(8192, r, 8) for r in (0, 7)] == [(0, 1024), (7168, 8192)]


def test_two_rank_allgather_of_the_best_oracle_plan():
    sys.path.insert(0, ROOT)
    from tpl_amd import sharding
    n, world = 23, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    records = [_oracle_record(i) for i in range(n)]
    serial = sharding.local_best(records)
    for rank, lo, hi, glob in results:
        assert np.array_equal(glob, serial), rank
    # the record really is a plan: id, validity measure and value of the winning trajectory
    T = GRID[3]
    traj = serial[3:].reshape(T, 7)
    assert serial[0] == np.max(traj[1:-1, 6]) and serial[1] == traj[0, 5] and 0 <= serial[2] < n
    assert len({(r[0], r[1]) for r in records}) > n // 2          # the scenarios are really different plans
    assert all(not sharding.better(r, serial) for r in records)
    spans = sorted((lo, hi) for _, lo, hi, _ in results)
    assert spans[0][0] == 0 and spans[-1][1] == n and spans[0][1] == spans[1][0]


def test_eight_rank_allgather_over_uneven_blocks():
    """BASELINE configs[4] rehearsed on the CPU (no 8-GPU node was ever available to this repository): EIGHT gloo ranks
    shard 8 190 scenario ids -- blocks of 1 024 and 1 023, the uneven case 8 192 never shows --, every rank plans its block
    with the CPU oracle, reduces it to its best-plan record, ONE all-gather, and every rank must hold the same global
    argmin with ranks_seen == 8.  The first run on hardware should fail, if it fails, on RCCL and nothing else."""
    sys.path.insert(0, ROOT)
    from tpl_amd import sharding
    n, world = 8190, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    spans = [(lo, hi) for _, lo, hi, _, _, _ in results]
    assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert sorted(hi - lo for lo, hi in spans) == [1023] * 2 + [1024] * 6
    assert all(seen == world for _, _, _, seen, _, _ in results)
    glob = results[0][5]
    for rank, lo, hi, _, best, g in results:
        assert np.array_equal(g, glob), rank                                   # identical argmin on every rank
        assert lo <= best[2] < hi                                              # every rank contributed a plan of its own block
    assert np.array_equal(glob, sharding.local_best([r[4] for r in results]))  # = the host-side reduction of the eight records
    wid = int(glob[2])
    assert np.array_equal(glob, _oracle_record(wid))                           # the record IS that scenario's plan
    rng = np.random.default_rng(0)
    for sid in rng.integers(0, n, 150):                                        # ... and no sampled scenario beats it
        assert not sharding.better(_oracle_record(int(sid)), glob), sid


def test_order_is_constraint_then_cost_then_id():
    from tpl_amd import sharding
    T = 4
    base = np.zeros((T, 7), dtype=np.float32)

    def rec(constr, cost, sid):
        tr = base.copy()
        tr[1, 6] = constr
        tr[0, 5] = cost
        return sharding.plan_record(tr, sid)
    assert sharding.better(rec(0.0, 9.0, 5), rec(0.1, 1.0, 0))          # a feasible plan beats any cheaper violating one
    assert sharding.better(rec(0.0, 1.0, 5), rec(0.0, 2.0, 0))
    assert sharding.better(rec(0.0, 1.0, 3), rec(0.0, 1.0, 4))          # ties: lowest id, like the planner's lowest action
    assert not sharding.better(rec(0.0, 1.0, 3), rec(0.0, 1.0, 3))
    # first and last state do not count towards the validity measure (poly_lat_dp_lon_planner.py:99-101)
    tr = base.copy()
    tr[0, 6] = 7.0
    tr[-1, 6] = 7.0
    assert sharding.plan_record(tr, 0)[0] == 0.0


def test_device_record_layout_decodes_to_the_host_record():
    """TplDpPlanRecord + TplLonState[t_steps] as the device writes it (include/tplhip.h) -> plan_record layout."""
    from tpl_amd import sharding
    T = GRID[3]
    want = _oracle_record(3)
    traj = want[3:].reshape(T, 7).astype(np.float32)
    raw = np.zeros(16 + 28 * T, dtype=np.uint8)
    raw[0:4] = np.array([want[0]], np.float32).view(np.uint8)
    raw[4:8] = np.array([want[1]], np.float32).view(np.uint8)
    raw[8:12] = np.array([3], np.int32).view(np.uint8)
    raw[12:16] = np.array([T], np.int32).view(np.uint8)
    raw[16:] = traj.view(np.uint8).ravel()
    got = sharding.record_from_device((float(raw[0:4].view(np.float32)[0]), float(raw[4:8].view(np.float32)[0]),
                                       int(raw[8:12].view(np.int32)[0]), raw[16:].view(np.float32).reshape(T, 7)))
    assert np.array_equal(got, want)


def test_single_process_skips_the_collective():
    from tpl_amd import sharding
    rec = _oracle_record(3)
    assert sharding.allgather_best(rec, None) is rec


def test_bench_launcher_starts_its_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` (no torchrun): the parent starts two child ranks before touching a GPU; on a box
    without one every rank refuses to run (no CPU fallback) and the launcher reports failure instead of a line."""
    import subprocess
    if torch.cuda.device_count() > 0 and os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the launcher is exercised by the bench itself")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-secondary", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert b"needs a GPU" in p.stderr or b"No HIP GPUs are available" in p.stderr or b"no usable HIP device" in p.stderr


def test_mc_scenarios_block_builder_equals_the_per_scenario_maps():
    """bench.py builds the occupancy maps of a rank's Monte-Carlo scenarios with one broadcast per block of scenarios
    (eight ranks set up on one host together); scenario by scenario they are benchkit.scenarios.occ_map of the same
    obstacles, and a scenario id gives the same scenario whatever block it is built in."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    from benchkit import scenarios as sc
    ep = sc.env_params(t_steps=7, s_steps=101, l_steps=11)
    occs, inits = bench.mc_scenarios(ep, 5, 9)
    for i in range(9):
        rng = np.random.default_rng(77000 + 5 + i)
        k = int(rng.integers(2, 7))
        obst = tuple((float(a), float(b), 4.0) for a, b in zip(rng.uniform(30.0, 180.0, k), rng.uniform(0.0, 10.0, k)))
        assert np.array_equal(occs[i], sc.occ_map(ep, obst)), i
        v0, a0 = float(rng.uniform(2.0, 14.0)), float(rng.uniform(-1.0, 1.0))
        assert (inits[i].v, inits[i].a) == (np.float32(v0), np.float32(a0))
    again, _ = bench.mc_scenarios(ep, 9, 3)
    assert np.array_equal(again, occs[4:7])


def test_headline_scenarios_differ_between_ranks():
    import bench
    o0, i0 = bench.rank_scenario(0)
    o3, i3 = bench.rank_scenario(3)
    from benchkit import scenarios as sc
    assert o0 == sc.OBSTACLES_OVERTAKE_TRAFFIC and (i0.s, i0.v, i0.a) == (0.0, 8.0, 0.0)
    assert o3 != o0 and i3.v == 9.5


def test_bench_deck_deals_every_pair_to_every_lane():
    """bench.py's timed region deals every plan its own (obstacle set, start state) pair by plan index: four sets and seven states
    with coprime periods, the two planners in flight walking the deck with different offsets -- each lane sees all 28 pairs, two
    consecutive plans of a lane never share set or state, and the deck's maps / states are really different (rank-dependent)."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    from benchkit import scenarios as sc
    for lanes in (1, 2):
        seen = [set() for _ in range(lanes)]
        last = [None] * lanes
        for i in range(28 * lanes):
            io, ist = bench.deal(i, lanes)
            assert 0 <= io < 4 and 0 <= ist < 7
            lane = i % lanes
            if last[lane] is not None:
                assert last[lane][0] != io and last[lane][1] != ist
            last[lane] = (io, ist)
            seen[lane].add((io, ist))
        assert all(len(s) == 28 for s in seen)
    ep = sc.env_params(t_steps=6, s_steps=101, l_steps=11)
    occs, starts = bench.scenario_deck(0, ep)
    assert len(occs) == 4 and len(starts) == 7
    assert len({o.tobytes() for o in occs}) == 4 and len({(s.s, s.v, s.a) for s in starts}) == 7
    occs3, starts3 = bench.scenario_deck(3, ep)
    assert all(a.tobytes() != b.tobytes() for a, b in zip(occs, occs3)) and starts3[0].v == 9.5
