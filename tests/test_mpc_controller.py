"""tpl_amd.control.ModelPredictiveController against the REFERENCE's controller class.  tests/golden/mpc_controller.npz
holds the controls, the ego trace and the predicted trajectories of closed loops in which
tpl.control.model_predictive_controller.ModelPredictiveController drove the reference's generated CPU solver
(generator: tests/golden/make_mpc_goldens.py; loops: tests/mpc_cases.py).

CPU: the mirror on the CPU restatement of the solver (oracle/ilqr_oracle.cpp) reproduces them -- controller logic and
solver pinned together.   GPU: the mirror on the HIP solver (its default) does."""
import os

import numpy as np
import pytest

import mpc_cases as cases
from tpl_amd.control import ModelPredictiveController, ModelPredictiveControllerTime

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "mpc_controller.npz"))
TOL = 2e-6          # closed loop over <= 120 ticks; solver parity itself is 1e-6 of scale (tests/test_ilqr_*.py)


def _check(name, factory):
    controls, trace, extra, con = cases.run(factory, name)
    assert np.allclose(controls, GOLD[f"{name}.controls"], rtol=0, atol=TOL), \
        f"{name}: controls differ by {np.max(np.abs(controls - GOLD[name + '.controls'])):.2e}"
    assert np.allclose(trace, GOLD[f"{name}.trace"], rtol=0, atol=TOL)
    for i, e in enumerate(extra):
        want = GOLD[f"{name}.con_traj_{i}"]
        assert e.shape == want.shape and np.allclose(e, want, rtol=0, atol=10 * TOL), f"{name}: predicted trajectory {i}"
    assert np.allclose(con.dead_time_trajectory, GOLD[f"{name}.dead_time_trajectory"], rtol=0, atol=TOL)
    state = np.array([con.lat_dist_to_traj, con.jerk, getattr(con, "cycle_time_estimate", 0.0), con.idle_comp_acc,
                      con.idle_comp_steer, len(con.ctrl_vars_history)])
    assert np.allclose(state, GOLD[f"{name}.state"], rtol=0, atol=10 * TOL)
    return con


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_controller_on_the_cpu_solver_matches_the_reference_loop(name):
    import oracle_ilqr
    con = _check(name, lambda shared, lock: ModelPredictiveController(shared, lock, solver=oracle_ilqr.OracleOptim(
        "trajectory_tracking_mpc")))
    assert ModelPredictiveController.update.runtime > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_controller_on_the_hip_solver_matches_the_reference_loop(name):
    from tpl_amd.optim.solver import Optim
    con = _check(name, lambda shared, lock: ModelPredictiveController(shared, lock))
    assert isinstance(con.opt, Optim)


@pytest.mark.parametrize("name", sorted(cases.TIME_CASES))
def test_time_indexed_controller_on_the_cpu_solver(name):
    import oracle_ilqr
    _check(name, lambda shared, lock: ModelPredictiveControllerTime(shared, lock, solver=oracle_ilqr.OracleOptim(
        "trajectory_tracking_mpc_time")))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.TIME_CASES))
def test_time_indexed_controller_on_the_hip_solver(name):
    _check(name, lambda shared, lock: ModelPredictiveControllerTime(shared, lock))


def test_short_trajectory_and_failed_resampling_keep_the_last_controls():
    import oracle_ilqr
    from types import SimpleNamespace as NS
    shared = NS()
    con = ModelPredictiveController(shared, cases.nolock, solver=oracle_ilqr.OracleOptim("trajectory_tracking_mpc"))
    con.controls = (0.3, -0.1)
    ego = cases.EgoSim()
    one = NS(time=np.zeros(1), s=np.zeros(1), x=np.zeros(1), y=np.zeros(1), orientation=np.zeros(1), curvature=np.zeros(1),
             velocity=np.zeros(1))
    assert con.update(NS(t=0.0, vehicle=cases.vehicle_of(ego), trajectory=one))[0] == (0.3, -0.1)       # :113-114
    # time running backwards or jumping clears the dead-time history (:104-106)
    con.ctrl_vars_history = [(0.0, 0.0)] * 5
    con.last_update_time = 10.0
    con.update(NS(t=5.0, vehicle=cases.vehicle_of(ego), trajectory=one))
    assert con.ctrl_vars_history == []
