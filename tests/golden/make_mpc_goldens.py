#!/usr/bin/env python3
"""Generates tests/golden/mpc_controller.npz by running the REFERENCE's controller plugin
(tpl.control.model_predictive_controller.ModelPredictiveController on the reference's generated CPU solver, imported
from /root/reference in the build container) through the closed loops of tests/mpc_cases.py.

    CFLAGS="-DPyArray_MoveInto=PyArray_CopyInto" python tests/golden/make_mpc_goldens.py

Shims as in tests/golden/make_pvd_goldens.py (numba identity, objtoolbox merge/bundle, structstore unused, tplcpp's
project / resample = this repository's host routines); the shared-parameter object answers ``deepcopy()`` like a
structstore struct.  The ego model is benchkit.minisim.EgoSim on both sides."""
import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
os.environ.setdefault("CFLAGS", "-DPyArray_MoveInto=PyArray_CopyInto")


class _Shared:
    """Attribute store whose `params` comes back with deepcopy()."""

    def __setattr__(self, k, v):
        if not hasattr(v, "deepcopy"):
            v.deepcopy = lambda v=v: copy.deepcopy(v)
        object.__setattr__(self, k, v)


def main():
    import make_pvd_goldens as shims
    shims.install_shims()
    # objtoolbox.merge writes the attributes the DESTINATION has (the time-indexed problem has no p_phi_ref_dot_diff)
    sys.modules["objtoolbox"].merge = lambda dst, src: [setattr(dst, k, getattr(src, k)) for k in
                                                        (getattr(dst, "__slots__", None) or vars(dst)) if hasattr(src, k)] and None
    sys.path.insert(0, "/root/reference/library")
    from make_ilqr_goldens import reference_factory
    import mpc_cases as cases
    from tpl.optim import optimizers as ref_opts
    ref_opts.trajectory_tracking_mpc = reference_factory("trajectory_tracking_mpc")
    from tpl.control.model_predictive_controller import ModelPredictiveController
    ref_opts.trajectory_tracking_mpc_time = reference_factory("trajectory_tracking_mpc_time")
    from tpl.control.model_predictive_controller_time import ModelPredictiveControllerTime
    out = {}
    for name in list(cases.CASES) + list(cases.TIME_CASES):
        Controller = ModelPredictiveController if name in cases.CASES else ModelPredictiveControllerTime

        def factory(shared, lock):
            # cases.run sets its overrides on shared.params: hand it the controller's own parameter object
            st = _Shared()
            con = Controller(st, lock)
            shared.params = st.params
            return con
        controls, trace, extra, con = cases.run(factory, name)
        out[f"{name}.controls"], out[f"{name}.trace"] = controls, trace
        for i, e in enumerate(extra):
            out[f"{name}.con_traj_{i}"] = e
        out[f"{name}.dead_time_trajectory"] = con.dead_time_trajectory
        out[f"{name}.state"] = np.array([con.lat_dist_to_traj, con.jerk, getattr(con, "cycle_time_estimate", 0.0),
                                         con.idle_comp_acc, con.idle_comp_steer, len(con.ctrl_vars_history)])
        print(name, controls[[0, 1, -1]].round(4).tolist(), trace[-1].round(3).tolist())
    path = os.path.join(HERE, "mpc_controller.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
