"""The planner plugins that drive the DP engines (tpl.planning.dyn_prog): ``DpEnv`` feeds the occupancy pipeline from
tpl's environment state; ``PolyLatDpLonPlanner`` chains the lateral lattice and the (s,v,a) planner into the trajectory
tplsim's controllers track; ``DpLatLonPlanner`` does the same around the (s,ds,l) planner.  Host Python like the
reference's plugins; every engine call goes to libtplhip.so."""
from .env_feed import DpEnv, Params as DpEnvParams  # noqa: F401
from .plugins import (DpLatLonParams, DpLatLonPlanner, EnvironmentState, PolyLatDpLonParams,  # noqa: F401
                      PolyLatDpLonPlanner)
