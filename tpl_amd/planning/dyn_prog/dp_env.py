"""``DpEnv`` (library/tpl/planning/dyn_prog/dp_env.py): the planner-side owner of the DP environment.  Every cycle it
turns tpl's environment state -- local map, predicted objects -- into the reference line and the prediction triangle
soups of ``DynProgEnvironment`` and runs the occupancy pipeline.

``engines`` is the module the engine classes come from: ``tpl_amd`` (the HIP engines) by default; the CPU test-suite
hands in an object with the same surface."""
import copy
import time

import numpy as np

from ...store import Bundle, merge
from ..geometry import gen_prediction_geometry, project
from ..utils import rampify_profile


def _engines(engines):
    if engines is not None:
        return engines
    import tpl_amd
    return tpl_amd


class Params:
    """dp_env.py:12-31"""

    def __init__(self, engines=None):
        self.write_debug_data = True
        self.dead_time = 0.0
        self.a_lat_max = 2.5
        self.a_max_v_ref = 3.0
        self.a_min_v_ref = -3.0
        self.j_max_v_ref = 1.5
        self.j_min_v_ref = -1.5
        self.t_dist_on_map = 0.5
        self.t_dist_crossing = 3.0
        self.cpp = _engines(engines).DynProgEnvParams()

    def deepcopy(self):
        return copy.deepcopy(self)


class DpEnv:

    def __init__(self, shared, lock_shared, engines=None):
        self._engines = _engines(engines)
        self.shared = shared
        self.lock_shared = lock_shared
        self.ref_line = None
        self.ref_proj = None
        self.last_update_time = 0.0
        self.dt_start = None
        self.ref_line_shift = 0.0
        self.ref_line_step_size = 0.0
        self.cpp_env = self._engines.DynProgEnvironment()
        self.runtime_environment = 0.0
        with self.lock_shared():                                            # :52-55
            self.shared.params.env = Params(self._engines)
            self.shared.debug.env = Bundle()

    def update_params(self, env):
        with self.lock_shared():                                            # :57-75
            params = self.shared.params.env
            params.cpp.dilation = np.sqrt(2.0) * env.vehicle_state.width * 0.5
            dt_update = env.t - self.last_update_time
            if self.dt_start is None:
                self.dt_start = params.cpp.dt
            else:
                self.dt_start = (self.dt_start - dt_update) % params.cpp.dt
            params.cpp.dt_start = self.dt_start
            sh_params = params.deepcopy()
        params = Params(self._engines)
        merge(params, sh_params)
        return params

    def update_reference_line(self, env, params):
        local_map = env.local_map
        if self.ref_line is not None:                                       # :81-86: how far the map window moved
            proj_prev_ref_start = project(self.ref_line[:, :2], local_map.path[0, :2])
            self.ref_line_shift = round(proj_prev_ref_start.arc_len / self.ref_line_step_size) * self.ref_line_step_size
        self.ref_line = np.zeros((len(local_map.path), 9))                  # :88-92
        self.ref_line[:, :6] = local_map.path
        self.ref_line[:, 6] = local_map.d_left
        self.ref_line[:, 7] = local_map.d_right
        self.ref_line_step_size = local_map.step_size_ref
        idxs_zero = self.ref_line[:, 5] < 1.0                               # :96-108: ramped speed limit
        self.ref_line[:, 5] = rampify_profile(None, None, self.ref_line[:, 5], params.a_min_v_ref, params.a_max_v_ref,
                                              params.j_min_v_ref, params.j_max_v_ref, 1.0, local_map.step_size_ref)[:, 0]
        self.ref_line[idxs_zero, 5] = 0.0
        for ip in local_map.intersection_paths:                             # :112-116 (returns, like the reference)
            if not ip.stop_proj.in_bounds:
                return
            self.ref_line[ip.stop_proj.end:ip.stop_proj.end + 10, 8] = 1.0
        params.cpp.l_min = np.floor(np.min(-self.ref_line[:, 7]))           # :120-121
        params.cpp.l_max = np.ceil(np.max(self.ref_line[:, 6]))

    def update_environment(self, env, params):
        start = time.perf_counter()
        self.cpp_env.reinit_buffers(params.cpp)                             # :130-131
        self.cpp_env.set_ref_line(self.ref_line, self.ref_line_step_size)
        maps = {m.uuid: m for m in env.get_relevant_maps()}
        for obj in env.predicted:                                           # :135-168
            for pred in obj.predictions:
                try:
                    m = maps[pred.uuid_assoc_map]
                except KeyError:
                    continue
                on_local_map = m.name == "local_map_behind"
                ts = np.array([0.0, *(params.cpp.dt_start + pred.states[:-1, 0])])
                ts += params.dead_time
                if on_local_map:
                    sweep_length = params.t_dist_on_map
                else:
                    sweep_length = params.t_dist_crossing
                    if env.vehicle_state.v > 20.0 or (obj.a is not None and obj.a > 1.0):
                        sweep_length = 4.0
                    elif env.vehicle_state.v > 15.0:
                        sweep_length = 3.0
                    elif env.vehicle_state.v > 10.0:
                        sweep_length = 2.0
                geom = gen_prediction_geometry(pred.states, obj.hull, m.path[:, :2], ts, station_step_size=5.0,
                                               expansion_rate=0.0, sweep_length=sweep_length)
                geom[:, 2] -= params.dead_time
                self.cpp_env.insert_geometry(geom, obj.stationary)
        self.cpp_env.update()
        self.runtime_environment = (time.perf_counter() - start) * 1000.0

    def write_debug_data(self):
        with self.lock_shared():                                            # :174-189
            dbg = self.shared.debug.env
            dbg.runtime_environment = self.runtime_environment
            dbg.ref_line = self.ref_line
            dbg.occ_map_cart = self.cpp_env.get_occ_map_cartesian()
            dbg.occ_map = np.vstack(np.moveaxis(self.cpp_env.get_occ_map(), 2, 0))
            dbg.dist_map_lon = np.vstack(np.moveaxis(self.cpp_env.get_dist_map_lon(), 2, 0))
            dbg.dir_dist_map = self.cpp_env.get_dist_map_dir(0)

    def update(self, env):
        params = self.update_params(env)                                    # :191-201
        self.update_reference_line(env, params)
        self.update_environment(env, params)
        if params.write_debug_data:
            self.write_debug_data()
        self.last_update_time = env.t
