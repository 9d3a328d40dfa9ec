"""``tpl.planning.path_vel_decomp``'s three solver users on the MI355X solvers: same class names, parameters,
attributes and call signatures (``update(env, params)`` / ``update(path, env, params)``)."""
from .path_optim import PathOptim
from .path_smoothing import PathSmoothing
from .velocity_optim import TimeConstr, VelocityOptim
from .path_vel_decomp_planner import Params as PathVelDecompParams, PathVelDecompPlanner  # noqa: E402

__all__ = ["PathOptim", "PathSmoothing", "VelocityOptim", "TimeConstr", "PathVelDecompPlanner", "PathVelDecompParams"]
