"""Scenario batches of the DP (s,v,a) planner (tplhip_dp_lon_scenarios_*): every scenario of a batched
launch must be bit-identical to the CPU oracle's plan of the same scenario -- value tables, trajectory -- and the
device-side reduction must pick the plan `sharding.local_best` picks from the oracle's plans."""
import numpy as np
import pytest

import oracle_dp as O
import tpl_amd
from benchkit import scenarios as sc
from tpl_amd import sharding

pytestmark = pytest.mark.gpu


def _scenarios(n, ep, seed):
    rng = np.random.default_rng(seed)
    occs, inits = [], []
    for i in range(n):
        obst = sc.random_obstacles(seed=100 * seed + i, k=2 + (i % 5))
        occs.append(sc.occ_map(ep, obst if i % 4 else ()))        # every fourth scenario is an empty road
        inits.append(sc.initial_state(0.0, float(rng.uniform(2.0, 14.0)), float(rng.uniform(-1.0, 1.0))))
    return np.stack(occs), inits


def _traj_np(traj):
    return O.states_to_np([s._c for s in traj.states])


CASES = [
    # S, V, A, T, n scenarios, chunk, per-scenario paths
    pytest.param(64, 16, 9, 5, 1, 1, False, id="B1-two-rows-per-lane"),
    pytest.param(128, 32, 12, 5, 3, 3, False, id="B3-s128-v32"),
    pytest.param(128, 64, 32, 4, 8, 8, False, id="B8-config2-grid"),
    pytest.param(201, 37, 20, 6, 8, 3, False, id="B8-in-chunks-of-3-reference-default-grid"),     # odd v_steps, ragged rows
    pytest.param(96, 16, 10, 5, 4, 4, True, id="B4-own-path-per-scenario"),
    pytest.param(96, 17, 10, 5, 3, 3, "step0.3", id="B3-path-step-0.3"),      # a divisor without a proven two-float form: Newton form or IEEE
    pytest.param(96, 17, 10, 5, 3, 3, "general", id="B3-general-sweep"),       # TPLHIP_DEBUG=no_v3: the batched form of k_backward
    # rule R3 under another blend order: the batched form of k_backward3o (and of the general sweep's eight-tap path)
    pytest.param(128, 32, 12, 5, 3, 3, "order1", id="B3-blend-order-s-v-a"),
    pytest.param(96, 17, 10, 5, 4, 2, "order2", id="B4-eight-corner-sum-in-chunks"),
    pytest.param(96, 17, 10, 5, 3, 3, "order3-general", id="B3-wide-sum-general-sweep"),
]


@pytest.fixture
def request_cleanup():
    todo = []
    yield todo
    for f in todo:
        f()


@pytest.mark.parametrize("S,V,A,T,n,chunk,own_paths", CASES)
def test_batched_scenarios_bit_exact(S, V, A, T, n, chunk, own_paths, monkeypatch, request_cleanup):
    p = sc.lon_params(S, V, A, T)
    want_kernel = "k_backward3v"
    if own_paths == "step0.3":
        p.path_step_size, p.path_steps, own_paths = 0.3, 667, False
    elif own_paths == "general":
        monkeypatch.setenv("TPLHIP_DEBUG", "no_v3")
        own_paths, want_kernel = False, "k_backward"
    order = 0
    if isinstance(own_paths, str) and own_paths.startswith("order"):
        order = int(own_paths[5])
        want_kernel = "k_backward3o"
        if own_paths.endswith("general"):
            monkeypatch.setenv("TPLHIP_DEBUG", "no_v3")
            want_kernel = "k_backward"
        own_paths = False
        O.set_rule(order=order)
        request_cleanup.append(lambda: O.set_rule())
    ep = sc.env_params(t_steps=T, s_steps=201, l_steps=31)
    occs, inits = _scenarios(n, ep, seed=S + n)
    if own_paths:
        paths = np.stack([sc.straight_path(p.path_steps, p.path_step_size, l=0.2 * (i - 1)) for i in range(n)])
        for i in range(n):
            paths[i, :, 5] += 0.25 * i                              # another speed limit profile per scenario
    else:
        paths = sc.straight_path(p.path_steps, p.path_step_size)[None]
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    if order:
        dp.set_blend_order(order)
    dp.upload_scenarios(occs, ep, inits, paths, chunk=chunk)
    assert dp.info()["scenario_kernel"] == want_kernel, dp.info()
    cp, cep = O.cparams(p), O.cparams(ep)
    want_trajs, records = [], []
    for first in range(0, n, chunk):
        count = min(chunk, n - first)
        dp.plan_scenarios(first, count, id_base=1000, reset_best=(first == 0))
        for i in range(first, first + count):
            path_i = paths[i if own_paths else 0]
            tables, otrajs, _ = O.update(cp, cep, path_i, occs[i], [inits[i]])
            for t in range(1, T):
                got = dp.copy_scenario_table(i, t)
                assert np.array_equal(got.view(np.uint32), tables[t].view(np.uint32)), (i, t)
            want_trajs.append(otrajs[0])
            records.append(sharding.plan_record(otrajs[0], 1000 + i))
    got = dp.download_scenarios()
    for i in range(n):
        assert np.array_equal(_traj_np(got[i]).view(np.uint32), want_trajs[i].view(np.uint32)), i
    # device-side reduction == host-side reduction of the oracle's plans
    best = sharding.local_best(records)
    constr, cost, rid, states = dp.best_record()
    assert (constr, cost, rid) == (np.float32(best[0]), np.float32(best[1]), int(best[2]))
    assert np.array_equal(states.view(np.uint32), want_trajs[rid - 1000].view(np.uint32))
    assert sharding.record_from_device(dp.best_record())[:3].tolist() == [best[0], best[1], best[2]]
    # a scenario planned alone through the reference-shaped entry point gives the same plan
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(ep)
    env.set_occ_map(occs[n - 1])
    single = dp.update(inits[n - 1], paths[(n - 1) if own_paths else 0], env)
    assert np.array_equal(_traj_np(single).view(np.uint32), want_trajs[n - 1].view(np.uint32))


def test_scenario_api_errors():
    p = sc.lon_params(64, 16, 9, 5)
    ep = sc.env_params(t_steps=5, s_steps=201, l_steps=31)
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(p)
    occs, inits = _scenarios(2, ep, seed=1)
    path = sc.straight_path(p.path_steps, p.path_step_size)
    with pytest.raises(RuntimeError, match="no scenarios uploaded"):
        dp._sc_n = 2
        dp.plan_scenarios(0, 1)
    with pytest.raises(ValueError, match="incorrect shape"):
        dp.upload_scenarios(occs[:, :-1], ep, inits, path)
    dp.upload_scenarios(occs, ep, inits, path, chunk=1)
    with pytest.raises(ValueError, match="does not fit"):
        dp.plan_scenarios(0, 2)                                    # more than one chunk per launch
    dp.plan_scenarios(1, 1)
    with pytest.raises(RuntimeError, match="not among those of the last launch"):
        dp.copy_scenario_table(0, 1)
    # a size change of the grid drops the scenarios (their tables are sized by it)
    dp.reinit_buffers(sc.lon_params(65, 16, 9, 5))
    with pytest.raises(RuntimeError, match="no scenarios uploaded"):
        dp.plan_scenarios(0, 1)
