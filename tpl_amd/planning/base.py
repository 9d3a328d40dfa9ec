"""What a planner plugin returns and the interface it implements (tpl.planning.Trajectory / BasePlanner)."""
import numpy as np

_SERIES = ("time", "s", "x", "y", "orientation", "curvature", "velocity", "acceleration")


class Trajectory:
    """Eight equally long series plus the emergency flag; a fresh one holds a single zero sample."""

    def __init__(self):
        for name in _SERIES:
            setattr(self, name, np.zeros((1,)))
        self.emergency = False


class BasePlanner:
    def __init__(self, state, params):
        pass

    def update(self, env):
        return Trajectory()
