"""BASELINE configs[0] ("tplsim default scenario, OCP solver N=20, single vehicle") reduced to the hot path:
benchkit/minisim.py -- the simulator's ego model with actuator dead time (core.py:91-135) under the controller's
per-tick solver use (model_predictive_controller.py:96-237).  CPU: the loop closed with the oracle solver converges
to the lane.  GPU: the same loop closed with the HIP solver applies the same controls, tick for tick."""
import numpy as np
import pytest

import oracle_ilqr as OI
from benchkit import minisim


def test_ego_model_dead_time_and_limits():
    ego = minisim.EgoSim(x=0.0, y=0.0, yaw=0.0, v=5.0, acc_dead_time=0.03, steer_dead_time=0.0)
    ego.control_acc, ego.control_steer = 1.0, 0.2
    seen = []
    t = 0.0
    for _ in range(6):
        ego.step(t, 0.01)
        t += 0.01
        seen.append((ego.a, ego.steer_angle))
    # steering without dead time acts at once; the acceleration request shows up once the buffer spans the dead time
    assert all(s == 0.2 for _, s in seen)
    assert seen[0][0] == 1.0 and seen[-1][0] == 1.0      # constant request: the delayed value is the same value
    ego2 = minisim.EgoSim(v=5.0, acc_dead_time=0.03, steer_dead_time=0.0)
    t, acc = 0.0, []
    for k in range(8):
        ego2.control_acc = float(k)                      # a ramp: what is applied lags by the dead time
        ego2.step(t, 0.01)
        t += 0.01
        acc.append(ego2.a)
    assert acc[0] == 0.0 and acc[-1] < 7.0 and all(b >= a for a, b in zip(acc, acc[1:]))
    # speed and steering limits (core.py:129-133)
    ego3 = minisim.EgoSim(v=39.99, acc_dead_time=0.0, steer_dead_time=0.0)
    ego3.control_acc, ego3.control_steer = 10.0, 2.0
    ego3.step(0.0, 0.01)
    assert ego3.v == 40.0 and ego3.steer_angle == 0.7


def test_closed_loop_with_the_cpu_solver_converges_to_the_lane():
    opt = OI.OracleOptim("trajectory_tracking_mpc")
    opt.horizon = 20
    controls, trace, con = minisim.closed_loop(opt, ticks=1000)
    assert abs(trace[-1, 1]) < 0.01 and abs(trace[-1, 2]) < 0.005          # on the lane, heading along it
    assert abs(trace[-1, 3] - 8.0) < 0.05                                   # at the plan's speed
    assert np.all(np.abs(controls[:, 0]) <= 3.0) and np.all(np.abs(controls[:, 1]) <= 0.7)
    assert np.max(np.abs(trace[:, 1])) < 0.45                               # never further out than it started
    assert len(con.ctrl_vars_history) == 100                                # the dead-time history is capped (:204-205)


@pytest.mark.gpu
def test_hip_solver_applies_the_same_controls_tick_for_tick():
    from tpl_amd.optim import genopt, optimizers as o
    make = genopt.build(o.config_trajectory_tracking_mpc(), "trajectory_tracking_mpc")
    opt = make()
    opt.horizon = 20
    got_c, got_t, _ = minisim.closed_loop(opt, ticks=1000)
    ref = OI.OracleOptim("trajectory_tracking_mpc")
    ref.horizon = 20
    want_c, want_t, _ = minisim.closed_loop(ref, ticks=1000)
    # two INDEPENDENT closed loops (each solver steers its own vehicle): any solver difference would compound
    assert np.max(np.abs(got_c - want_c)) < 1e-6, np.max(np.abs(got_c - want_c))
    assert np.max(np.abs(got_t - want_t)) < 1e-6
    assert abs(got_t[-1, 1]) < 0.01 and abs(got_t[-1, 3] - 8.0) < 0.05
