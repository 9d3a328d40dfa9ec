"""tpl's path/velocity-decomposition planner on the MI355X solvers: the two lateral stages (``lateral.py``), the speed
stage (``speed.py``) and the planner plugin that chains them (``planner.py``) -- same class names, shared-parameter
blocks, attributes and call signatures as ``tpl.planning.path_vel_decomp``."""
from .lateral import OptimParams, PathOptim, PathSmoothing, SmoothingParams, rampify_profile as rampify_lateral_profile
from .planner import Params as PathVelDecompParams, PathVelDecompPlanner
from .speed import SpeedParams, TimeConstr, VelocityOptim

__all__ = ["PathOptim", "PathSmoothing", "VelocityOptim", "TimeConstr", "PathVelDecompPlanner", "PathVelDecompParams",
           "OptimParams", "SmoothingParams", "SpeedParams", "rampify_lateral_profile"]
