"""tpl_amd/store.py: the StructStoreRegistry adapter (library/tpl/util.py:310-332) and the reader of objtoolbox's
state.json layout (a small fixture in that layout is committed under tests/golden/state/)."""
import copy
import os

import numpy as np

from tpl_amd import store
from tpl_amd.planning.path_vel_decomp import OptimParams

HERE = os.path.dirname(os.path.abspath(__file__))


def test_state_json_reader():
    st = store.load_state(os.path.join(HERE, "golden", "state"))
    assert st["active_planner"] == "path_vel_decomp_planner"
    po = st["path_vel_decomp_planner"].params.path_optim
    assert po.horizon == 250 and po.cost_func.w_a_d == 200.0
    assert getattr(st["path_vel_decomp_planner"], "__class_tag__") == "objtoolbox.utils.bundle"
    assert st["window_pos"].dtype == np.float32 and list(st["window_pos"]) == [0.0, 37.0]
    assert np.array_equal(st["control_points"], np.arange(6.0).reshape(3, 2)) and st["missing_points"] is None
    assert st["cars"][0].uuid == "u1" and st["plain"] == {"a": 1}
    # the json file itself works too, and parameter objects of the glue take the file's values (otb.merge)
    again = store.load_state(os.path.join(HERE, "golden", "state", "state.json"))
    p = OptimParams()
    store.merge(p, again["path_vel_decomp_planner"].params.path_optim)
    assert p.cost_func.w_a_d == 200.0 and p.evasion_sharpness == 20.0 and p.lateral_min_gap == 2.0    # untouched default


def test_registry_hands_out_one_store_per_path():
    a = store.StructStoreRegistry.get("/tpl_test_planning")
    b = store.StructStoreRegistry.get("tpl_test_planning")
    assert a is b
    with a.lock():
        a.params = store.Bundle(horizon=20)
        a.counter = 3
    assert a.revalidate() in (True, None)
    c = a.deepcopy()
    c.params.horizon = 30
    assert a.params.horizon == 20 and c.counter == 3
    d = copy.deepcopy(a)
    assert d is not a and d.params.horizon == 20
    assert store.StructStoreRegistry.get("tpl_test_control") is not a
