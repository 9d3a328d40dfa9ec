"""The controller plugins that drive the batched iLQR solver (tpl.control's two tracking MPCs): HP-2's callers."""
from .tracking_mpc import (BaseController, ModelPredictiveController, ModelPredictiveControllerTime, Params,  # noqa: F401
                           TimeParams)
