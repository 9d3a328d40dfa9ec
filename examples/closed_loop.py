#!/usr/bin/env python3
"""A mini tplsim on the MI355X engines: planner plugin -> controller plugin -> the simulator's ego model, in closed loop.

    python examples/closed_loop.py [poly_lat_dp_lon | dp_lat_lon] [seconds]

The planner (tpl_amd.planning.dyn_prog: occupancy pipeline + lattice + DP (s,v,a), or DP (s,ds,l)) runs at 10 Hz on the
synthetic road of benchkit/drive.py with three predicted objects; the trajectory-tracking MPC
(tpl_amd.control.ModelPredictiveController on the HIP iLQR solver) runs at 100 Hz; the ego vehicle is the kinematic
single-track model with actuator dead time of tpl's simulator (tpl_amd/minisim.EgoSim).  Needs an MI355X."""
import os
import sys
import time
from contextlib import contextmanager
from types import SimpleNamespace as NS

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchkit import drive                                                   # noqa: E402
from tpl_amd.control import ModelPredictiveController                       # noqa: E402
from benchkit.minisim import EgoSim                                          # noqa: E402
from tpl_amd.planning.dyn_prog import DpLatLonPlanner, PolyLatDpLonPlanner  # noqa: E402
from tpl_amd.store import Bundle                                            # noqa: E402


@contextmanager
def lock():
    yield


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "poly_lat_dp_lon"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    planner = (DpLatLonPlanner if kind == "dp_lat_lon" else PolyLatDpLonPlanner)(Bundle(), lock)
    con_shared = Bundle()
    controller = ModelPredictiveController(con_shared, lock)
    # the controller weights tpl ships for its default vehicle (data/params/control/default/state.json)
    for k, v in dict(pd=10.0, pv=5.0, pdelta=0.0, min_pdelta_dot=0.1, pdelta_dot=0.0, min_p_phi_dot=0.0, p_phi_dot=0.05,
                     p_phi=1000.0, p_phi_ref_dot_diff=0.01, pa=2.0, pj=0.5).items():
        setattr(con_shared.params.cost_function, k, v)
    con_shared.params.horizon = 20
    d = drive.Drive(v_ego=9.0)
    sh = d.cycle(0)
    v0 = sh.vehicle_state
    ego = EgoSim(x=v0.x, y=v0.y, yaw=v0.phi, v=v0.v, wheel_base=v0.wheel_base)
    t_plan, t_ctrl = [], []
    traj = None
    n_ticks = int(seconds * 100)
    for tick in range(n_ticks):
        t = d.t + tick * 0.01
        if tick % 10 == 0:                                                   # planner cycle: the vehicle as it is NOW
            sh = d.cycle(tick // 10, follow=(ego.x, ego.y, ego.yaw, ego.v, ego.a))
            t0 = time.perf_counter()
            traj = planner.update(sh)
            t_plan.append((time.perf_counter() - t0) * 1e3)
        if traj is None or traj.emergency:
            acc, steer = -3.0, 0.0
        else:
            veh = NS(x=ego.x, y=ego.y, phi=ego.yaw, delta=ego.steer_angle, v=ego.v, a=ego.a, pitch=0.0,
                     wheel_base=ego.wheel_base, delta_max=ego.max_steer_angle, dead_time_steer=ego.steer_dead_time)
            t0 = time.perf_counter()
            (acc, steer), _ = controller.update(NS(t=t, vehicle=veh, trajectory=traj))
            t_ctrl.append((time.perf_counter() - t0) * 1e3)
        ego.control_acc, ego.control_steer = acc, steer
        for _ in range(2):
            ego.step(t, 0.005)
            t += 0.005
        if tick % 100 == 99 and traj is not None and not traj.emergency:
            # the vehicle's centre of gravity in the frame of the plan's point for NOW (the plan is in CoG coordinates)
            i = int(np.clip(np.searchsorted(traj.time, t), 0, len(traj.time) - 1))
            dx = ego.x + np.cos(ego.yaw) * ego.wheel_base * 0.5 - traj.x[i]
            dy = ego.y + np.sin(ego.yaw) * ego.wheel_base * 0.5 - traj.y[i]
            c, s_ = np.cos(traj.orientation[i]), np.sin(traj.orientation[i])
            print(f"t = {tick / 100 + 0.01:4.1f} s   v = {ego.v:5.2f} m/s (plan {traj.velocity[i]:5.2f})   cross-track {-s_ * dx + c * dy:+6.3f} m   "
                  f"along-track {c * dx + s_ * dy:+6.2f} m   planner p50 {np.percentile(t_plan, 50):5.2f} ms   "
                  f"controller p50 {np.percentile(t_ctrl, 50):5.2f} ms")
    print(f"{kind}: {len(t_plan)} planner cycles, {len(t_ctrl)} controller ticks; final speed {ego.v:.2f} m/s")


if __name__ == "__main__":
    main()
