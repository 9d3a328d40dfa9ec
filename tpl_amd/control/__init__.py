"""The controller plugin that drives the batched iLQR solver (library/tpl/control/): HP-2's caller in tplsim."""
from .model_predictive_controller import BaseController, ModelPredictiveController, Params  # noqa: F401
from .model_predictive_controller_time import ModelPredictiveControllerTime  # noqa: F401
