"""tpl_amd -- MI355X-native hot path for tpl (DP (s,v,a) planner + genopt iLQR).

``tpl_amd.dyn_prog``  mirrors the HP-1 slice of the reference's ``tplcpp`` module,
``tpl_amd.optim``     mirrors ``tpl.optim`` (genopt / optimizers) for HP-2.
Everything computes in ``libtplhip.so``; importing this package without the
built library raises ImportError.
"""

from . import _capi

_capi.load()

from .dyn_prog import (  # noqa: E402
    DynProgEnvParams,
    DynProgEnvironment,
    DynProgLonPlanner,
    DynProgLonPlannerParams,
    LonState,
    LonTraj,
    cuda_get_device_count,
    cuda_set_device,
    lon_dynamics,
)

from .dyn_prog import RefLine, RefLinePoint  # noqa: E402
from .lat_lon import (  # noqa: E402
    DynProgLatLonPlanner,
    DynProgLatLonPlannerParams,
    LatDistances,
    LatLonCartState,
    LatLonCartTraj,
    LatLonState,
    LatLonTraj,
)
from .poly_lat import (  # noqa: E402
    PolyLatPlanner,
    PolyLatPlannerParams,
    PolyLatTraj,
    PolyLatTrajPoint,
    PolyQuintic,
)

from .planning.geometry import Projection, gen_prediction_geometry, project, resample  # noqa: E402
from .planning.geometry import lerp_cpp as lerp  # noqa: E402  (tplcpp.lerp, utils.cpp:931-990)
from .planning.geometry import convex_hull, intersect_polygons, point_in_polygon, smooth_path  # noqa: E402

__all__ = [
    "Projection", "gen_prediction_geometry", "lerp", "project", "resample", "convex_hull", "intersect_polygons",
    "point_in_polygon", "smooth_path",
    "DynProgLatLonPlanner", "DynProgLatLonPlannerParams", "LatLonCartState", "LatLonCartTraj", "LatLonState",
    "LatLonTraj",
    "PolyLatPlanner", "PolyLatPlannerParams", "PolyLatTraj", "PolyLatTrajPoint", "PolyQuintic",
    "DynProgEnvParams", "DynProgEnvironment", "DynProgLonPlanner", "DynProgLonPlannerParams",
    "LonState", "LonTraj", "RefLine", "RefLinePoint", "LatDistances", "cuda_get_device_count", "cuda_set_device", "lon_dynamics",
]
