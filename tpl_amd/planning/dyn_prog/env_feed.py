"""``DpEnv``: the planner-side owner of the DP environment (behaviour of tpl.planning.dyn_prog.dp_env.DpEnv,
dp_env.py:34-201).  Every cycle it turns tpl's environment state into the two inputs of ``DynProgEnvironment`` -- the
reference line (local map + lane bounds + a driveable speed limit + intersection flags) and one prediction triangle
soup per predicted object -- and runs the occupancy pipeline.

``engines`` is the module the engine classes come from: ``tpl_amd`` (the HIP engines) unless a test hands in an object
with the same surface."""
import time

import numpy as np

from ...store import Bundle
from .._kit import ParamBlock, StagePhase, snapshot
from ..geometry import gen_prediction_geometry, project
from ..utils import rampify_profile

# reference-line columns (RefLinePoint): the local map's six path columns, then these
D_LEFT, D_RIGHT, SEMANTIC, V_MAX = 6, 7, 8, 5
STATION_STEP = 5.0                  # length of one swept box segment of a prediction, metres
CONFLICT_ZONE_POINTS = 10           # reference-line points flagged after an intersection's stop line


def _engines(engines):
    if engines is not None:
        return engines
    import tpl_amd
    return tpl_amd


class Params(ParamBlock):
    DEFAULTS = dict(write_debug_data=True, dead_time=0.0, a_lat_max=2.5, a_max_v_ref=3.0, a_min_v_ref=-3.0,
                    j_max_v_ref=1.5, j_min_v_ref=-1.5, t_dist_on_map=0.5, t_dist_crossing=3.0,
                    cpp=lambda engines: _engines(engines).DynProgEnvParams())


def _sweep_seconds(on_own_lane, params, ego_speed, obj_acc):
    """How far ahead of a predicted object its occupancy is swept, in seconds of its own motion: short for traffic on
    the ego lane, longer for crossing traffic and growing with the ego speed (or when the object accelerates)."""
    if on_own_lane:
        return params.t_dist_on_map
    if ego_speed > 20.0 or (obj_acc is not None and obj_acc > 1.0):
        return 4.0
    for threshold, seconds in ((15.0, 3.0), (10.0, 2.0)):
        if ego_speed > threshold:
            return seconds
    return params.t_dist_crossing


class DpEnv:

    def __init__(self, shared, lock_shared, engines=None):
        self._engines = _engines(engines)
        self.shared, self.lock_shared = shared, lock_shared
        self.ref_line = None
        self.ref_proj = None
        self.ref_line_shift = 0.0
        self.ref_line_step_size = 0.0
        self.last_update_time = 0.0
        self._phase = StagePhase()
        self.cpp_env = self._engines.DynProgEnvironment()
        self.runtime_environment = 0.0
        with self.lock_shared():
            self.shared.params.env = Params(self._engines)
            self.shared.debug.env = Bundle()

    @property
    def dt_start(self):
        return self._phase.value

    def update(self, env):
        params = self.update_params(env)
        self.update_reference_line(env, params)
        self.update_environment(env, params)
        if params.write_debug_data:
            self.write_debug_data()
        self.last_update_time = env.t

    def update_params(self, env):
        def per_cycle(block):
            block.cpp.dilation = np.sqrt(2.0) * env.vehicle_state.width * 0.5        # half the diagonal of a square of that width
            block.cpp.dt_start = self._phase.advance(env.t - self.last_update_time, block.cpp.dt)
        return snapshot(self.lock_shared, self.shared.params.env, Params, self._engines, per_cycle)

    def update_reference_line(self, env, params):
        lm = env.local_map
        if self.ref_line is not None:
            # how far the map window moved since the last cycle, in whole reference steps
            moved = project(self.ref_line[:, :2], lm.path[0, :2]).arc_len
            self.ref_line_shift = round(moved / self.ref_line_step_size) * self.ref_line_step_size
        line = np.zeros((len(lm.path), 9))
        line[:, :6], line[:, D_LEFT], line[:, D_RIGHT] = lm.path, lm.d_left, lm.d_right
        self.ref_line, self.ref_line_step_size = line, lm.step_size_ref
        # a speed limit the vehicle can actually follow: ramped with bounded acceleration and jerk, standstill kept
        standstill = line[:, V_MAX] < 1.0
        line[:, V_MAX] = rampify_profile(None, None, line[:, V_MAX], params.a_min_v_ref, params.a_max_v_ref,
                                         params.j_min_v_ref, params.j_max_v_ref, 1.0, lm.step_size_ref)[:, 0]
        line[standstill, V_MAX] = 0.0
        for crossing in lm.intersection_paths:
            stop = crossing.stop_proj
            if not stop.in_bounds:
                return          # as the source does: the first stop line off the map ends the method (lateral range stays)
            line[stop.end:stop.end + CONFLICT_ZONE_POINTS, SEMANTIC] = 1.0
        params.cpp.l_min = np.floor(np.min(-line[:, D_RIGHT]))      # the lateral grid covers the whole lane
        params.cpp.l_max = np.ceil(np.max(line[:, D_LEFT]))

    def update_environment(self, env, params):
        started = time.perf_counter()
        cpp = self.cpp_env
        cpp.reinit_buffers(params.cpp)
        cpp.set_ref_line(self.ref_line, self.ref_line_step_size)
        known_maps = {m.uuid: m for m in env.get_relevant_maps()}
        for obj in env.predicted:
            for pred in obj.predictions:
                road = known_maps.get(pred.uuid_assoc_map)
                if road is None:
                    continue
                # slice times: now, then the planner's stage boundaries; the dead time shifts them while the geometry
                # is generated and is taken out again afterwards
                slice_times = np.concatenate(([0.0], params.cpp.dt_start + pred.states[:-1, 0])) + params.dead_time
                sweep = _sweep_seconds(road.name == "local_map_behind", params, env.vehicle_state.v, obj.a)
                soup = gen_prediction_geometry(pred.states, obj.hull, road.path[:, :2], slice_times,
                                               station_step_size=STATION_STEP, expansion_rate=0.0, sweep_length=sweep)
                soup[:, 2] -= params.dead_time
                cpp.insert_geometry(soup, obj.stationary)
        cpp.update()
        self.runtime_environment = (time.perf_counter() - started) * 1000.0

    def write_debug_data(self):
        cpp = self.cpp_env
        # the Frenet tables are published with their time slices below each other ([l, s, t(, k)] -> [t * l, s(, k)])
        if hasattr(cpp, "get_occ_map_slices"):
            occ_slices, dist_slices = cpp.get_occ_map_slices, cpp.get_dist_map_lon_slices      # transposed on the device
        else:
            occ_slices = lambda: np.vstack(np.moveaxis(cpp.get_occ_map(), 2, 0))
            dist_slices = lambda: np.vstack(np.moveaxis(cpp.get_dist_map_lon(), 2, 0))
        with self.lock_shared():
            dbg = self.shared.debug.env
            dbg.runtime_environment = self.runtime_environment
            dbg.ref_line = self.ref_line
            dbg.occ_map_cart = cpp.get_occ_map_cartesian()
            dbg.occ_map = occ_slices()
            dbg.dist_map_lon = dist_slices()
            dbg.dir_dist_map = cpp.get_dist_map_dir(0)
