"""``PolyLatDpLonPlanner`` (library/tpl/planning/dyn_prog/poly_lat_dp_lon_planner.py): the planner plugin around the
lateral quintic lattice and the DP (s,v,a) planner -- HP-1's caller in tpl.  Per cycle: environment update, shift of
the previous plan onto the moved map window, the replan decision, lateral plan -> path -> speed plan, assembly of the
trajectory the controllers track.  Engine calls go to ``engines`` (default ``tpl_amd``: libtplhip.so)."""
import copy
import time

import numpy as np

from ...store import Bundle, merge
from ..base import BasePlanner, Trajectory
from ..geometry import curv_to_vel_profile, lerp, project
from ..utils import rampify_profile
from .dp_env import DpEnv, _engines


class Params:
    """poly_lat_dp_lon_planner.py:28-46"""

    def __init__(self, engines=None):
        e = _engines(engines)
        self.write_debug_data = True
        self.update_always = False
        self.replan_time_step = 0.1
        self.dead_time = 0.0
        self.dist_path_fix_min = 5.0
        self.dist_path_fix = 1.0
        self.d_reinit = 2.0
        self.cpp_lat = e.PolyLatPlannerParams()
        self.cpp_lon = e.DynProgLonPlannerParams()

    def deepcopy(self):
        return copy.deepcopy(self)


class EnvironmentState:
    """The slice of tpl.environment.EnvironmentState the planner copies out of the shared state (:413-423)."""

    def __init__(self):
        self.t = 0.0
        self.reset_counter = 0
        self.vehicle_state = None
        self.local_map = None
        self.local_map_behind = None
        self.selected_map = None
        self.tracks = Bundle()              # sensor name -> list of tracked objects
        self.cpms = []
        self.predicted = []
        self.man_time_cons = []
        self.man_vel_cons = []

    def get_all_tracks(self):
        """env_state.py:141-148: every tracked object of every source, plus the infrastructure's."""
        groups = self.tracks.values() if isinstance(self.tracks, dict) else (
            [self.tracks] if isinstance(self.tracks, list) else
            [v for k, v in vars(self.tracks).items() if not k.startswith("_")])
        objs = []
        for g in groups:
            objs += copy.deepcopy(g)
        objs += copy.deepcopy(self.cpms)
        return objs

    def get_relevant_maps(self):
        """env_state.py:125-139: the map behind/around the vehicle and the intersecting paths whose stop point lies
        on the local map."""
        if self.local_map is None:
            return []
        maps = [self.local_map_behind]
        for ip in self.local_map.intersection_paths:
            if project(self.local_map.path[:, :2], ip.stop_pos).in_bounds:
                maps.append(ip.map_segment)
        return maps


class PolyLatDpLonPlanner(BasePlanner):

    def __init__(self, shared, lock_shared, engines=None):
        np.seterr(divide="ignore", invalid="ignore")
        e = self._engines = _engines(engines)
        self.shared = shared
        self.lock_shared = lock_shared
        self.reset_counter = 0
        self.invalid_counter = 0
        self.emergency_counter = 0
        self.last_update_time = -1.0
        self.last_replan_time = -1.0
        self.dt_start = None
        self.state_reinit_msg = ""
        self.ref_proj = None
        self.traj_lat = None
        self.path = None
        self.traj_lon = None
        self.traj_point_prev = None
        self.trajectory = Trajectory()
        self.trajectory_np = None
        self.poly_lat_start = e.PolyLatTrajPoint()
        self.dp_lon_start = e.LonState()
        self.poly_lat = e.PolyLatPlanner()
        self.dp_lon = e.DynProgLonPlanner()
        self.debug = False
        self.runtime_dp = 0.0
        with self.lock_shared():                                                    # :79-85
            self.shared.params = Bundle()
            self.shared.params.planner = Params(e)
            self.shared.debug = Bundle()
            self.shared.debug.planner = Bundle()
        self.dp_env = DpEnv(shared, lock_shared, e)
        self.env = EnvironmentState()

    def is_traj_valid(self, traj):
        return all(s.constr < 0.1 for s in traj.states[1:-1])

    def update_params(self, env):
        veh = env.vehicle_state
        with self.lock_shared():                                                    # :98-127
            params = self.shared.params.planner
            length_veh = veh.rear_axis_to_front + veh.rear_axis_to_rear
            params.cpp_lat.length_veh = length_veh
            params.cpp_lat.width_veh = env.vehicle_state.width
            params.cpp_lon.length_veh = length_veh
            params.cpp_lon.width_veh = env.vehicle_state.width
            dt_update = env.t - self.last_update_time
            if self.dt_start is None:
                self.dt_start = params.cpp_lon.dt
            else:
                self.dt_start = (self.dt_start - dt_update) % params.cpp_lon.dt
            if self.dt_start == 0.0:
                self.dt_start = params.cpp_lon.dt
            params.cpp_lon.dt_start = self.dt_start
            sh_params = params.deepcopy()
        params = Params(self._engines)
        merge(params, sh_params)
        return params

    def update_planner(self, env, params):
        start = time.perf_counter()
        self.poly_lat.reinit_buffers(params.cpp_lat)                               # :138-142
        new_traj_lat = self.poly_lat.update(self.poly_lat_start, self.dp_env.cpp_env)
        if self.traj_lat is None:
            self.traj_lat = new_traj_lat
        else:
            self.traj_lat.insert_after_station(self.poly_lat_start.s, new_traj_lat)
        # the lateral trajectory resampled into the speed planner's path (:150-155): PathState columns
        # x, y, s, l, k, v_max, distance
        self.path = self.traj_lat.lerp(np.arange(0.0, params.cpp_lon.path_steps * params.cpp_lon.path_step_size,
                                                 params.cpp_lon.path_step_size))
        self.path = self.path[:, [7, 8, 5, 1, 11, 6, 10]]
        self.path[:, 5] = curv_to_vel_profile(self.path[:, 4], self.path[:, 5], params.cpp_lat.a_lat_abs_max)   # :157-168
        self.path[:, 5] = rampify_profile(None, None, self.path[:, 5], params.cpp_lon.a_min, params.cpp_lon.a_max,
                                          params.cpp_lon.j_min, params.cpp_lon.j_max, 1.0, 1.0)[:, 0]
        # where the previous trajectory is at the start of the new plan, on the new path (:172-177)
        self.traj_point_prev = lerp(env.t + params.dead_time, self.trajectory_np[:, 0], self.trajectory_np[:, 2:4])
        path_proj = project(self.path[:, :2], self.traj_point_prev)
        self.dp_lon_start.s = path_proj.arc_len
        self.dp_lon.reinit_buffers(params.cpp_lon)                                 # :179-183
        self.traj_lon = self.dp_lon.update(self.dp_lon_start, self.path, self.dp_env.cpp_env)
        self.runtime_dp = time.perf_counter() - start
        self.last_replan_time = env.t

    def update_trajectory(self, env, params):
        ts = np.arange(0.0, (params.cpp_lon.t_steps - 1) * params.cpp_lon.dt, 0.1)                      # :188-206
        lon_states = self.traj_lon.state(ts)
        lat_states = self.traj_lat.lerp(lon_states[:, 0])
        traj_np = np.zeros((len(ts), 8))
        traj_np[:, 0] = ts + env.t + params.dead_time
        traj_np[:, 1] = lat_states[:, 5]
        traj_np[:, 2] = lat_states[:, 7]
        traj_np[:, 3] = lat_states[:, 8]
        traj_np[:, 4] = lon_states[:, 1]
        traj_np[:, 5] = lon_states[:, 2]
        traj_np[:, 6] = lat_states[:, 9]
        traj_np[:, 7] = lat_states[:, 11]
        ts = np.arange(env.t, env.t + params.dead_time, 0.1)                                           # :208-222
        if len(ts) > 0:
            traj_dead_time = lerp(ts, self.trajectory_np[:, 0], self.trajectory_np)
            traj_dead_time = traj_dead_time.reshape(len(ts), -1)
            traj_dead_time[:, 6] = lerp(ts, self.trajectory_np[:, 0], self.trajectory_np[:, 6], angle=True)
            traj_np = np.concatenate((traj_dead_time, traj_np), axis=0)
        if not self.is_traj_valid(self.traj_lon):                                                      # :224-232
            self.invalid_counter += 1
        else:
            self.invalid_counter = 0
        if self.invalid_counter > 10:
            self.invalid_counter = 0
            self.emergency_counter = 50
        self.emergency_counter = max(0, self.emergency_counter - 1)
        traj = Trajectory()                                                                            # :234-248
        traj.emergency = self.emergency_counter > 0
        if not traj.emergency:
            traj.time = traj_np[:, 0]
            traj.s = traj_np[:, 1]
            traj.x = traj_np[:, 2]
            traj.y = traj_np[:, 3]
            traj.velocity = traj_np[:, 4]
            traj.acceleration = traj_np[:, 5]
            traj.orientation = traj_np[:, 6]
            traj.curvature = traj_np[:, 7]
            self.trajectory_np = traj_np
        self.trajectory = traj

    def reset_initial_state(self, env, params):
        veh = env.vehicle_state
        self.traj_lon = None                                                                           # :256-260
        self.traj_lat = None
        self.path = None
        self.trajectory_np = None
        self.trajectory = None
        ts = np.arange(0.0, 10.0, 0.1)                                                                 # :264-273: constant velocity
        self.trajectory_np = np.zeros((len(ts), 8))
        self.trajectory_np[:, 0] = ts + env.t
        self.trajectory_np[:, 1] = ts * veh.v
        self.trajectory_np[:, 2] = veh.x + np.cos(veh.phi) * (ts * veh.v + veh.wheel_base * 0.5)
        self.trajectory_np[:, 3] = veh.y + np.sin(veh.phi) * (ts * veh.v + veh.wheel_base * 0.5)
        self.trajectory_np[:, 4] = veh.v
        self.trajectory_np[:, 5] = 0.0
        self.trajectory_np[:, 6] = veh.phi
        self.trajectory_np[:, 7] = 0.0
        ref_proj = project(self.dp_env.ref_line[:, :2], [veh.x, veh.y])                                # :277-286
        self.poly_lat_start = self._engines.PolyLatTrajPoint()
        self.poly_lat_start.t = 0.0
        self.poly_lat_start.s = 0.0
        self.poly_lat_start.l = ref_proj.distance
        self.poly_lat_start.dl = np.tan(veh.phi - ref_proj.angle)
        self.poly_lat_start.ddl = 0.0
        self.poly_lat_start.dddl = 0.0
        self.poly_lat_start.v = veh.v
        self.dp_lon_start.t = 0.0                                                                      # :290-294
        self.dp_lon_start.s = 0.0
        self.dp_lon_start.v = veh.v
        self.dp_lon_start.a = min(params.cpp_lon.a_max, max(0.0, veh.a))
        self.dp_lon_start.j = 0.0

    def shift_trajectory(self, env, params):
        if self.traj_lon is None:
            return
        shift = env.local_map.shift_idx_start_ref * env.local_map.step_size_ref                        # :303-312
        if hasattr(self.traj_lat, "shift_stations"):
            self.traj_lat.shift_stations(shift)             # the two statements below on the C array
        else:
            pts = self.traj_lat.points
            for p in pts:
                p.s -= shift
            self.traj_lat.points = [p for p in pts if p.s >= 0.0]
        self.poly_lat_start = self.traj_lat.lerp(params.dist_path_fix_min + params.dist_path_fix * env.vehicle_state.v)
        self.poly_lat_start.v = env.vehicle_state.v
        self.trajectory_np[:, 1] -= shift                                                              # :316
        dt_update = env.t - self.last_update_time                                                      # :320-331
        for s in self.traj_lon.states:
            s.t -= dt_update
        self.traj_lon.states = [self.traj_lon.state(0.0)] + [s for s in self.traj_lon.states if s.t > 0.0]
        self.dp_lon_start = self.traj_lon.states[0]
        self.dp_lon_start.a = min(params.cpp_lon.a_max, max(params.cpp_lon.a_min, self.dp_lon_start.a))

    def check_replan(self, env, params):
        veh = env.vehicle_state
        self.ref_proj = project(env.local_map.path[:, :2], [veh.x, veh.y])
        if not veh.automated:                                                                          # :340-344
            self.state_reinit_msg = ""
            self.reset_initial_state(env, params)
            if env.t - self.last_replan_time >= 1.0:
                return True
        reset_required = self.reset_counter != env.reset_counter                                       # :346-352
        self.reset_counter = env.reset_counter
        if self.traj_lon is None or reset_required or self.trajectory.emergency:
            self.state_reinit_msg = ""
            self.reset_initial_state(env, params)
            return True
        x_cog = veh.x + np.cos(veh.phi) * veh.wheel_base * 0.5                                          # :356-364
        y_cog = veh.y + np.sin(veh.phi) * veh.wheel_base * 0.5
        d_traj = np.linalg.norm([self.trajectory.x[0] - x_cog, self.trajectory.y[0] - y_cog])
        if abs(d_traj) > params.d_reinit:
            self.state_reinit_msg = ("Warning: Planner reinit, distance to trajectory too high"
                                     + f"#{int(time.time() / 5) * 5}")
            self.reset_initial_state(env, params)
            return True
        if len(self.traj_lon.states) < params.cpp_lon.t_steps:                                         # :368-369
            return True
        if env.t - self.last_replan_time > params.replan_time_step:                                    # :373-374
            return True
        self.dp_lon.reinit_buffers(params.cpp_lon)                                                     # :378-386
        self.traj_lon = self.dp_lon.reeval_traj(self.traj_lon, self.path, self.dp_env.cpp_env)
        if not self.is_traj_valid(self.traj_lon):
            return True
        return False

    def write_debug_data(self, t, params, veh):
        if not params.write_debug_data:
            return
        if self.traj_lon is None or self.traj_lat is None:
            return
        with self.lock_shared():                                                                       # :398-408
            dbg = self.shared.debug.planner
            dbg.traj_point_prev = self.traj_point_prev
            dbg.traj_lon = self.traj_lon.as_numpy()
            dbg.traj_lat = self.traj_lat.as_numpy()
            dbg.path = self.path
            dbg.runtime_dp = self.runtime_dp

    def update(self, sh_env):
        env = self.env
        with sh_env.lock():                                                                            # :414-423
            env.t = sh_env.t
            env.reset_counter = sh_env.reset_counter
            env.vehicle_state = copy.deepcopy(sh_env.vehicle_state)
            env.local_map = copy.deepcopy(sh_env.local_map)
            env.local_map_behind = copy.deepcopy(sh_env.local_map_behind)
            env.selected_map = copy.deepcopy(sh_env.selected_map)
            env.tracks = copy.deepcopy(sh_env.tracks)
            env.predicted = copy.deepcopy(sh_env.predicted)
            env.man_time_cons = copy.deepcopy(sh_env.man_time_cons)
        params = self.update_params(env)
        update_needed = True
        if env.local_map is None:
            update_needed = False
        if env.t == self.last_update_time and not params.update_always:                                # :432-435
            time.sleep(0.001)
            update_needed = False
        if env.t < self.last_update_time:                                                              # :437-439
            self.last_update_time = 0.0
        if update_needed:                                                                              # :441-449
            self.dp_env.update(env)
            self.shift_trajectory(env, params)
            if self.check_replan(env, params):
                self.update_planner(env, params)
            self.update_trajectory(env, params)
            self.last_update_time = env.t
        self.write_debug_data(env.t, params, env.vehicle_state)
        return self.trajectory
