"""Building blocks shared by the planner and controller plugins (own design; the plugins reproduce the behaviour of
tpl's classes, this module is how they are put together here).

* ``ParamBlock``       parameter objects declared as a table of defaults (nested blocks allowed), deep-copyable, with the
                       ``deepcopy()`` a structstore proxy offers.
* ``snapshot``         "lock, deepcopy, merge into a fresh block": how every plugin reads its shared parameters.
* ``StagePhase``       the offset of the first planning stage inside the planner's time grid, carried from cycle to cycle.
* ``EnvSnapshot``      the planner's private copy of tpl's shared environment state.
* ``fill_trajectory``  [n, 8] rows (time, s, x, y, velocity, acceleration, orientation, curvature) -> Trajectory.
"""
import copy

import numpy as np

from ..store import Bundle, merge
from .base import Trajectory
from .geometry import project


_ATOMS = frozenset((int, float, bool, str, type(None)))


class ParamBlock:
    """Subclasses set ``DEFAULTS = {name: value | ParamBlock subclass | callable(engines) -> value}``."""
    DEFAULTS = {}

    def __init__(self, engines=None):
        for name, default in type(self).DEFAULTS.items():
            if isinstance(default, type) and issubclass(default, ParamBlock):
                value = default(engines)
            elif callable(default):
                value = default(engines)
            else:
                value = copy.copy(default)
            setattr(self, name, value)

    def deepcopy(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        """Attribute by attribute, without the pickle protocol copy.deepcopy falls back to for plain objects (a
        plugin copies its parameter tree twice per cycle)."""
        twin = object.__new__(type(self))
        memo[id(self)] = twin
        fill = twin.__dict__
        for name, value in self.__dict__.items():
            fill[name] = value if type(value) in _ATOMS else copy.deepcopy(value, memo)
        return twin


def snapshot(lock, shared_block, cls, engines=None, before_copy=None):
    """The parameter object a cycle works with: under the lock, let ``before_copy`` write cycle-dependent values into
    the shared block, deep-copy it, and merge the copy into a fresh ``cls`` (so that attributes a user removed from the
    store fall back to their defaults)."""
    with lock():
        if before_copy is not None:
            before_copy(shared_block)
        frozen = shared_block.deepcopy()
    fresh = cls(engines) if issubclass(cls, ParamBlock) else cls()
    merge(fresh, frozen)
    return fresh


class StagePhase:
    """Time from "now" to the first stage boundary of a planner whose stages are ``dt`` apart and anchored in absolute
    time: starts at one full ``dt`` and shrinks by the time that passed since the previous cycle, modulo ``dt``."""

    def __init__(self, zero_means_full=False):
        self.value = None
        self.zero_means_full = zero_means_full

    def advance(self, elapsed, dt):
        self.value = dt if self.value is None else (self.value - elapsed) % dt
        if self.zero_means_full and self.value == 0.0:
            self.value = dt
        return self.value


class EnvSnapshot:
    """What a planner copies out of the shared environment at the start of a cycle, plus the two queries the plugins
    make on it (tpl/environment/env_state.py)."""
    COPIED = ("vehicle_state", "local_map", "local_map_behind", "selected_map", "tracks", "predicted", "man_time_cons")

    def __init__(self):
        self.t = 0.0
        self.reset_counter = 0
        for name in self.COPIED:
            setattr(self, name, None)
        self.tracks = Bundle()
        self.predicted, self.man_time_cons, self.man_vel_cons, self.cpms = [], [], [], []

    def take(self, shared_env):
        """Caller holds the environment's lock."""
        self.t, self.reset_counter = shared_env.t, shared_env.reset_counter
        for name in self.COPIED:
            setattr(self, name, copy.deepcopy(getattr(shared_env, name)))

    def get_all_tracks(self):
        """Every tracked object of every source, plus what the infrastructure reported."""
        src = self.tracks
        groups = src.values() if isinstance(src, dict) else ([src] if isinstance(src, list) else
                                                             [v for k, v in vars(src).items() if not k.startswith("_")])
        return [copy.deepcopy(o) for g in groups for o in g] + copy.deepcopy(self.cpms)

    def get_relevant_maps(self):
        """The map around the vehicle and every intersecting path whose stop point lies on the local map."""
        if self.local_map is None:
            return []
        centre = self.local_map.path[:, :2]
        return [self.local_map_behind] + [ip.map_segment for ip in self.local_map.intersection_paths
                                          if project(centre, ip.stop_pos).in_bounds]


_TRAJ_COLUMNS = ("time", "s", "x", "y", "velocity", "acceleration", "orientation", "curvature")


def fill_trajectory(rows, emergency=False):
    traj = Trajectory()
    traj.emergency = bool(emergency)
    if not emergency:
        for col, name in enumerate(_TRAJ_COLUMNS):
            setattr(traj, name, rows[:, col])
    return traj


def centre_of_gravity(veh):
    """The point half a wheel base ahead of the rear axle, where tpl's planners place the vehicle."""
    return np.array([veh.x + np.cos(veh.phi) * veh.wheel_base * 0.5, veh.y + np.sin(veh.phi) * veh.wheel_base * 0.5])
