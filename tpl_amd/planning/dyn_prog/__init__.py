"""The planner plugins that drive the DP engines (library/tpl/planning/dyn_prog/): ``DpEnv`` feeds the occupancy
pipeline from tpl's environment state, ``PolyLatDpLonPlanner`` chains the lateral lattice and the (s,v,a) planner into
the trajectory tplsim's controllers track, ``DpLatLonPlanner`` does the same around the (s,ds,l) planner.  Host Python like the reference's; every engine call goes to libtplhip.so."""
from .dp_env import DpEnv, Params as DpEnvParams  # noqa: F401
from .dp_lat_lon_planner import DpLatLonPlanner, Params as DpLatLonParams  # noqa: F401
from .poly_lat_dp_lon_planner import Params as PolyLatDpLonParams, PolyLatDpLonPlanner  # noqa: F401
