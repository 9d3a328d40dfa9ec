"""``tpl.planning.Trajectory`` / ``BasePlanner`` (library/tpl/planning/trajectory.py, base_planner.py): what a planner
plugin returns and the interface it implements."""
import numpy as np


class Trajectory:

    def __init__(self):
        self.time = np.zeros((1,))
        self.s = np.zeros((1,))
        self.x = np.zeros((1,))
        self.y = np.zeros((1,))
        self.orientation = np.zeros((1,))
        self.curvature = np.zeros((1,))
        self.velocity = np.zeros((1,))
        self.acceleration = np.zeros((1,))
        self.emergency = False


class BasePlanner:

    def __init__(self, state, params):
        pass

    def update(self, env):
        return Trajectory()
