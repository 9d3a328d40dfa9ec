"""The two planner plugins around the DP engines, built on one replanning skeleton.

* ``PolyLatDpLonPlanner`` -- lateral quintic lattice, then the DP (s,v,a) speed planner on the lattice's path: HP-1's
  caller in tpl (behaviour of planning/dyn_prog/poly_lat_dp_lon_planner.py:49-448).
* ``DpLatLonPlanner``     -- the DP (s,ds,l) planner with its smoothing pass (planning/dyn_prog/dp_lat_lon_planner.py:
  42-354), the method tpl's own simulation tests drive.

Both run the same cycle, which the reference writes out twice and which is ONE method here (``_ReplanningPlugin.update``):
private copy of the shared environment -> cycle parameters -> environment feed (``DpEnv``) -> carry the previous plan over
to the new time and map window -> decide whether to replan (an ordered rule list) -> plan -> assemble the trajectory ->
debug block.  The subclasses supply the rules, the carry-over, the planning step and the assembly.
Engine calls go to ``engines`` (default ``tpl_amd``: libtplhip.so)."""
import time

import numpy as np
from scipy.interpolate import interp1d

from ...store import Bundle
from .._kit import EnvSnapshot, ParamBlock, StagePhase, centre_of_gravity, fill_trajectory, snapshot
from ..base import BasePlanner, Trajectory
from ..geometry import curv_to_vel_profile, lerp, project
from ..utils import rampify_profile
from .env_feed import DpEnv, _engines

OUTPUT_STEP = 0.1                 # sampling of the trajectory handed to the controller, seconds
EMERGENCY_CYCLES = 50             # cycles an emergency stop is held once declared
EnvironmentState = EnvSnapshot    # the name tpl's plugins use for their private environment copy


class _ReplanningPlugin(BasePlanner):
    PARAM_CLASS = None

    def __init__(self, shared, lock_shared, engines=None):
        np.seterr(divide="ignore", invalid="ignore")
        self._engines = _engines(engines)
        self.shared, self.lock_shared = shared, lock_shared
        self.reset_counter = 0
        self.emergency_counter = 0
        self.last_update_time = self.last_replan_time = -1.0
        self.state_reinit_msg = ""
        self.ref_proj = None
        self.trajectory, self.trajectory_np = Trajectory(), None
        self.debug = False
        self.runtime_dp = 0.0
        self._build_engines(self._engines)
        with self.lock_shared():
            self.shared.params = Bundle()
            self.shared.params.planner = self.PARAM_CLASS(self._engines)
            self.shared.debug = Bundle()
            self.shared.debug.planner = Bundle()
        self.dp_env = DpEnv(shared, lock_shared, self._engines)
        self.env = EnvSnapshot()

    @property
    def dt_start(self):
        return self._phase.value

    # ---- the cycle ------------------------------------------------------------------------------------------
    def update(self, sh_env):
        env = self.env
        with sh_env.lock():
            env.take(sh_env)
        params = self.update_params(env)
        time_moved = env.t != self.last_update_time or params.update_always
        if not time_moved:
            time.sleep(0.001)                                  # the caller polls: do not spin
        if env.t < self.last_update_time:
            self.last_update_time = 0.0                        # the clock went backwards (simulation restart)
        if env.local_map is not None and time_moved:
            self.dp_env.update(env)
            self.shift_trajectory(env, params)
            self._plan(env, params, self.check_replan(env, params))
            self.update_trajectory(env, params)
            self.last_update_time = env.t
        self.write_debug_data(env.t, params, env.vehicle_state)
        return self.trajectory

    def update_params(self, env):
        veh = env.vehicle_state

        def per_cycle(block):
            self._vehicle_into(block, veh.rear_axis_to_front + veh.rear_axis_to_rear, veh.width)
            grid = self._grid_params(block)
            grid.dt_start = self._phase.advance(env.t - self.last_update_time, grid.dt)
        return snapshot(self.lock_shared, self.shared.params.planner, self.PARAM_CLASS, self._engines, per_cycle)

    def check_replan(self, env, params):
        """First matching rule wins.  A rule is (condition, reset the start state first?, replan?); a rule with
        replan None only has its side effects and the scan goes on."""
        veh = env.vehicle_state
        self.ref_proj = project(env.local_map.path[:, :2], [veh.x, veh.y])
        for condition, reset, verdict in self._replan_rules(env, params, veh):
            if not condition():
                continue
            if reset:
                self.reset_initial_state(env, params)
            if verdict is not None:
                return verdict
        return False

    def _off_the_plan(self, veh, params):
        gap = np.linalg.norm(np.array([self.trajectory.x[0], self.trajectory.y[0]]) - centre_of_gravity(veh))
        if abs(gap) <= params.d_reinit:
            return False
        self.state_reinit_msg = "Warning: Planner reinit, distance to trajectory too high" + f"#{int(time.time() / 5) * 5}"
        return True

    def _clear_message(self):
        self.state_reinit_msg = ""
        return True

    def _reset_requested(self, env):
        changed = self.reset_counter != env.reset_counter
        self.reset_counter = env.reset_counter
        return changed


# =================================================================================================================
class PolyLatDpLonParams(ParamBlock):
    DEFAULTS = dict(write_debug_data=True, update_always=False, replan_time_step=0.1, dead_time=0.0,
                    dist_path_fix_min=5.0, dist_path_fix=1.0, d_reinit=2.0,
                    cpp_lat=lambda e: _engines(e).PolyLatPlannerParams(),
                    cpp_lon=lambda e: _engines(e).DynProgLonPlannerParams())


class PolyLatDpLonPlanner(_ReplanningPlugin):
    PARAM_CLASS = PolyLatDpLonParams
    # columns of PolyLatTraj.lerp() rows that make up a PathState row (x, y, s, l, k, v_max, distance)
    PATH_FROM_LATERAL = [7, 8, 5, 1, 11, 6, 10]

    def _build_engines(self, e):
        self.invalid_counter = 0
        self._phase = StagePhase(zero_means_full=True)
        self.traj_lat = self.traj_lon = self.path = self.traj_point_prev = None
        self.poly_lat_start, self.dp_lon_start = e.PolyLatTrajPoint(), e.LonState()
        self.poly_lat, self.dp_lon = e.PolyLatPlanner(), e.DynProgLonPlanner()

    def _vehicle_into(self, block, length, width):
        for grid in (block.cpp_lat, block.cpp_lon):
            grid.length_veh, grid.width_veh = length, width

    def _grid_params(self, block):
        return block.cpp_lon

    def is_traj_valid(self, traj):
        return all(s.constr < 0.1 for s in traj.states[1:-1])

    # ---- carry the previous plan over ------------------------------------------------------------------------
    def shift_trajectory(self, env, params):
        if self.traj_lon is None:
            return
        moved = env.local_map.shift_idx_start_ref * env.local_map.step_size_ref
        self.traj_lat.shift_stations(moved)                     # stations relative to the new window; the past is dropped
        # the lateral plan is kept up to a look-ahead that grows with speed; the new lattice starts there
        self.poly_lat_start = self.traj_lat.lerp(params.dist_path_fix_min + params.dist_path_fix * env.vehicle_state.v)
        self.poly_lat_start.v = env.vehicle_state.v
        self.trajectory_np[:, 1] -= moved
        elapsed = env.t - self.last_update_time
        lon = self.traj_lon
        for st in lon.states:
            st.t -= elapsed
        lon.states = [lon.state(0.0)] + [st for st in lon.states if st.t > 0.0]
        self.dp_lon_start = lon.states[0]
        self.dp_lon_start.a = min(params.cpp_lon.a_max, max(params.cpp_lon.a_min, self.dp_lon_start.a))

    def reset_initial_state(self, env, params):
        """Forget the previous plan: a constant-velocity stand-in trajectory, the lattice and the speed planner start
        at the vehicle."""
        veh = env.vehicle_state
        self.traj_lon = self.traj_lat = self.path = self.trajectory = None
        ts = np.arange(0.0, 10.0, OUTPUT_STEP)
        travelled = ts * veh.v
        ahead = travelled + veh.wheel_base * 0.5
        rows = np.zeros((len(ts), 8))
        rows[:, 0], rows[:, 1] = ts + env.t, travelled
        rows[:, 2], rows[:, 3] = veh.x + np.cos(veh.phi) * ahead, veh.y + np.sin(veh.phi) * ahead
        rows[:, 4], rows[:, 6] = veh.v, veh.phi
        self.trajectory_np = rows
        on_ref = project(self.dp_env.ref_line[:, :2], [veh.x, veh.y])
        start = self.poly_lat_start = self._engines.PolyLatTrajPoint()
        start.l, start.dl, start.v = on_ref.distance, np.tan(veh.phi - on_ref.angle), veh.v
        lon = self.dp_lon_start
        lon.t = lon.s = lon.j = 0.0
        lon.v, lon.a = veh.v, min(params.cpp_lon.a_max, max(0.0, veh.a))

    # ---- the replan decision ---------------------------------------------------------------------------------
    def _replan_rules(self, env, params, veh):
        def stale_plan_is_invalid():
            self.dp_lon.reinit_buffers(params.cpp_lon)
            self.traj_lon = self.dp_lon.reeval_traj(self.traj_lon, self.path, self.dp_env.cpp_env)
            return not self.is_traj_valid(self.traj_lon)
        manual = not veh.automated
        return (
            (lambda: manual and self._clear_message(), True, None),                    # manual driving: always restart ...
            (lambda: manual and env.t - self.last_replan_time >= 1.0, False, True),     # ... and replan once a second
            (lambda: self._must_restart(env) and self._clear_message(), True, True),
            (lambda: self._off_the_plan(veh, params), True, True),
            (lambda: len(self.traj_lon.states) < params.cpp_lon.t_steps, False, True),  # plan consumed
            (lambda: env.t - self.last_replan_time > params.replan_time_step, False, True),
            (stale_plan_is_invalid, False, True),
        )

    def _must_restart(self, env):
        requested = self._reset_requested(env)                  # always consumed, whatever else holds
        return self.traj_lon is None or requested or self.trajectory.emergency

    # ---- planning ----------------------------------------------------------------------------------------------
    def _plan(self, env, params, replan):
        if replan:
            self.update_planner(env, params)

    def update_planner(self, env, params):
        started = time.perf_counter()
        lat, lon = params.cpp_lat, params.cpp_lon
        self.poly_lat.reinit_buffers(lat)
        fresh = self.poly_lat.update(self.poly_lat_start, self.dp_env.cpp_env)
        if self.traj_lat is None:
            self.traj_lat = fresh
        else:
            self.traj_lat.insert_after_station(self.poly_lat_start.s, fresh)     # keep the fixed part, append the new lattice
        stations = np.arange(0.0, lon.path_steps * lon.path_step_size, lon.path_step_size)
        path = self.traj_lat.lerp(stations)[:, self.PATH_FROM_LATERAL]
        # the path's speed limit: capped by lateral acceleration in curves, then made driveable for the speed planner
        path[:, 5] = curv_to_vel_profile(path[:, 4], path[:, 5], lat.a_lat_abs_max)
        path[:, 5] = rampify_profile(None, None, path[:, 5], lon.a_min, lon.a_max, lon.j_min, lon.j_max, 1.0, 1.0)[:, 0]
        self.path = path
        # the speed plan starts where the previous trajectory is at the start time, measured along the NEW path
        self.traj_point_prev = lerp(env.t + params.dead_time, self.trajectory_np[:, 0], self.trajectory_np[:, 2:4])
        self.dp_lon_start.s = project(path[:, :2], self.traj_point_prev).arc_len
        self.dp_lon.reinit_buffers(lon)
        self.traj_lon = self.dp_lon.update(self.dp_lon_start, path, self.dp_env.cpp_env)
        self.runtime_dp = time.perf_counter() - started
        self.last_replan_time = env.t

    # ---- trajectory assembly -----------------------------------------------------------------------------------
    def update_trajectory(self, env, params):
        lon = params.cpp_lon
        ts = np.arange(0.0, (lon.t_steps - 1) * lon.dt, OUTPUT_STEP)
        speed_plan = self.traj_lon.state(ts)                          # (s, v, a, j) over time
        lateral = self.traj_lat.lerp(speed_plan[:, 0])                # the lateral plan at those stations
        rows = np.column_stack([ts + env.t + params.dead_time, lateral[:, 5], lateral[:, 7], lateral[:, 8],
                                speed_plan[:, 1], speed_plan[:, 2], lateral[:, 9], lateral[:, 11]])
        bridge = np.arange(env.t, env.t + params.dead_time, OUTPUT_STEP)
        if len(bridge) > 0:                                           # until the dead time is over: the previous trajectory
            old = self.trajectory_np
            head = lerp(bridge, old[:, 0], old).reshape(len(bridge), -1)
            head[:, 6] = lerp(bridge, old[:, 0], old[:, 6], angle=True)
            rows = np.concatenate((head, rows), axis=0)
        self.invalid_counter = 0 if self.is_traj_valid(self.traj_lon) else self.invalid_counter + 1
        if self.invalid_counter > 10:                                 # invalid for a second: emergency stop
            self.invalid_counter, self.emergency_counter = 0, EMERGENCY_CYCLES
        self.emergency_counter = max(0, self.emergency_counter - 1)
        self.trajectory = fill_trajectory(rows, emergency=self.emergency_counter > 0)
        if not self.trajectory.emergency:
            self.trajectory_np = rows

    def write_debug_data(self, t, params, veh):
        if not params.write_debug_data or self.traj_lon is None or self.traj_lat is None:
            return
        with self.lock_shared():
            dbg = self.shared.debug.planner
            dbg.traj_point_prev = self.traj_point_prev
            dbg.traj_lon, dbg.traj_lat = self.traj_lon.as_numpy(), self.traj_lat.as_numpy()
            dbg.path = self.path
            dbg.runtime_dp = self.runtime_dp


# =================================================================================================================
class DpLatLonParams(ParamBlock):
    DEFAULTS = dict(write_debug_data=True, update_always=False, replan_time_step=0.1, dead_time=0.0, d_reinit=2.0,
                    cpp=lambda e: _engines(e).DynProgLatLonPlannerParams())


class DpLatLonPlanner(_ReplanningPlugin):
    PARAM_CLASS = DpLatLonParams

    def _build_engines(self, e):
        self._phase = StagePhase()
        self.ref_line, self.ref_line_shift, self.ref_line_step_size = None, 0.0, 0.0
        self.traj_dp = None
        self.dp_planner = e.DynProgLatLonPlanner()

    def _vehicle_into(self, block, length, width):
        block.cpp.length_veh, block.cpp.width_veh = length, width

    def _grid_params(self, block):
        return block.cpp

    def is_traj_valid(self, traj):
        return all(s.constr == 0 for s in traj.states[1:])

    def shift_trajectory(self, env, params):
        if self.traj_dp is None:
            return
        elapsed, moved = env.t - self.last_update_time, self.dp_env.ref_line_shift
        dp = self.traj_dp
        for st in dp.states:
            st.t -= elapsed
            st.s -= moved
        dp.states = [dp.state(0.0)] + [st for st in dp.states if st.t > 0.0]
        self.dp_planner.traj_dp = dp.copy()                      # assignment to the engine's member copies
        smooth = self.dp_planner.traj_smooth
        head = smooth.lerp(elapsed)                               # the smoothed plan continues from where it is now
        head.t = 0.0
        head.s -= moved
        smooth[0] = head

    def reset_initial_state(self, env, params):
        veh = env.vehicle_state
        on_ref = project(self.dp_env.ref_line[:, :2], centre_of_gravity(veh))
        start = self._engines.LatLonState()
        start.s = on_ref.arc_len + veh.v * params.dead_time
        start.ds, start.l = veh.v, self.ref_proj.distance
        self.dp_planner.traj_dp[0] = start
        self.dp_planner.traj_smooth[0] = start
        self.trajectory_np = None

    def _replan_rules(self, env, params, veh):
        def forget_emergency():
            self.emergency_counter = 0
            return self._clear_message()

        def stale_plan_is_invalid():
            self.dp_planner.reinit_buffers(params.cpp)
            self.traj_dp = self.dp_planner.reeval_traj(self.traj_dp, self.dp_env.cpp_env)
            return not self.is_traj_valid(self.traj_dp)
        return (
            (lambda: self.emergency_counter > 0 and self._clear_message(), True, True),
            (lambda: (not veh.automated) and env.t - self.last_replan_time >= 1.0 and forget_emergency(), True, True),
            (lambda: (self._reset_requested(env) | (self.traj_dp is None)) and forget_emergency(), True, True),     # both evaluated
            (lambda: self._off_the_plan(veh, params), True, True),
            (lambda: len(self.traj_dp.states) < params.cpp.t_steps, False, True),
            (lambda: env.t - self.last_replan_time > params.replan_time_step, False, True),
            (stale_plan_is_invalid, False, True),
        )

    def _plan(self, env, params, replan):
        self.update_planner(env, params, replan)

    def update_planner(self, env, params, replan):
        grid = params.cpp
        if env.man_time_cons:               # a manoeuvre's crossing-time window, relative to the plan's start
            where, earliest, latest = env.man_time_cons[0]
            offset = env.t + params.dead_time
            grid.t_st_min, grid.t_st_max = earliest - offset, latest - offset
            grid.s_st = project(self.dp_env.ref_line[:, :2], where).arc_len
        else:
            grid.t_st_min, grid.t_st_max, grid.s_st = 0.0, 1000.0, 0.0
        engine, scene = self.dp_planner, self.dp_env.cpp_env
        if replan:
            engine.reinit_buffers(grid)
            started = time.perf_counter()
            engine.update_traj_dp(scene)
            self.runtime_dp = (time.perf_counter() - started) * 1000.0
            self.traj_dp = engine.traj_dp.copy()
            self.last_replan_time = env.t
        engine.update_traj_smooth(scene)
        engine.update_traj_cart(scene.ref_line)

    def update_trajectory(self, env, params):
        rows = self.dp_planner.traj_smooth_cart.as_numpy()
        rows[:, 0] += env.t + params.dead_time
        if self.trajectory_np is None:
            self.trajectory_np = rows
        old = self.trajectory_np
        old[:, 6] = np.unwrap(old[:, 6], period=np.pi * 2.0)
        bridge = np.arange(env.t, env.t + params.dead_time, params.cpp.dt_smooth_traj)
        head = interp1d(old[:, 0], old, axis=0, fill_value="extrapolate")(bridge)
        rows = np.concatenate((head, rows), axis=0)
        self.emergency_counter = max(0, self.emergency_counter - 1) if self.is_traj_valid(self.traj_dp) else EMERGENCY_CYCLES
        self.trajectory = fill_trajectory(rows, emergency=self.emergency_counter > 0)
        self.trajectory_np = rows

    def write_debug_data(self, t, params, veh):
        engine = self.dp_planner
        with self.lock_shared():
            dbg = self.shared.debug.planner
            dbg.reinit_msg, dbg.runtime_dp = self.state_reinit_msg, self.runtime_dp
            if params.write_debug_data:
                for name in ("traj_dp", "traj_smooth", "traj_dp_cart", "traj_smooth_cart"):
                    setattr(dbg, name, getattr(engine, name).as_numpy())
