"""BASELINE configs[4] at its full per-GPU size: 1024 Monte-Carlo scenario rollouts on the 128x64x32 grid with 50 stages
(bench.py's `mc_rollouts`, same scenario generator, same chunking), reduced on the device to the best-plan record.  The
CPU oracle cannot plan 1024 such scenarios in a test, so it plans the ONE the device picked -- its trajectory must equal
the record's bit for bit -- plus a sample of other ids, none of which may order before the winner."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle_dp as O           # noqa: E402
import tpl_amd                  # noqa: E402
from benchkit import scenarios as sc   # noqa: E402
from tpl_amd import sharding    # noqa: E402

pytestmark = pytest.mark.gpu


def test_best_record_of_1024_rollouts_is_the_oracles_plan_of_that_scenario():
    import bench
    S, V, A, T = sc.GRIDS["config2"]
    assert (S, V, A, T) == (128, 64, 32, 50)
    n, chunk, id_base = bench.MC_PER_GPU, bench.MC_CHUNK, 0
    assert n == 1024
    p = sc.lon_params(S, V, A, T)
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    occs, inits = bench.mc_scenarios(ep, 0, n)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    dp.upload_scenarios(occs, ep, inits, path, chunk=chunk)
    assert dp.info()["scenario_kernel"] == "k_backward3v", dp.info()
    for first in range(0, n, chunk):
        dp.plan_scenarios(first, min(chunk, n - first), id_base=id_base, reset_best=(first == 0))
    constr, cost, rid, states = dp.best_record()
    assert 0 <= rid < n and states.shape == (T, 7)

    cp, cep = O.cparams(p), O.cparams(ep)
    _, otrajs, _ = O.update(cp, cep, path, occs[rid], [inits[rid]])
    assert np.array_equal(states.view(np.uint32), otrajs[0].view(np.uint32)), rid
    want = sharding.plan_record(otrajs[0], rid)
    assert (np.float32(want[0]), np.float32(want[1]), int(want[2])) == (constr, cost, rid)

    # a sample of the other scenarios: none may order before the winner under (constr, cost, id)
    rng = np.random.default_rng(5)
    for i in rng.choice(n, size=3, replace=False):
        if i == rid:
            continue
        _, ot, _ = O.update(cp, cep, path, occs[i], [inits[i]])
        other = sharding.plan_record(ot[0], int(i))
        assert (float(want[0]), float(want[1]), int(want[2])) < (float(other[0]), float(other[1]), int(other[2])), (i, other[:3], want[:3])
