#!/usr/bin/env python3
"""Writes tests/golden/genopt_surface.json: the attribute NAMES (and, for array properties, shapes and dtypes at
default construction) of the reference's generated solver type and its ``params`` object, read off the imported
reference module for every tpl problem.  Run here (needs /root/reference):

    CFLAGS="-DPyArray_MoveInto=PyArray_CopyInto" python tests/golden/make_genopt_surface.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
os.environ.setdefault("CFLAGS", "-DPyArray_MoveInto=PyArray_CopyInto")

from make_ilqr_goldens import reference_factory  # noqa: E402

MODELS = ["trajectory_tracking_mpc", "trajectory_tracking_mpc_time", "lateral_profile", "velocity_profile_space",
          "velocity_profile_time", "ref_line_smoother_k", "ref_line_smoother_dk"]


def describe(obj):
    out = {}
    for n in dir(obj):
        if n.startswith("__") and n not in ("__slots__", "__deepcopy__", "__getstate__"):
            continue
        try:
            v = getattr(obj, n)
        except Exception as e:                      # noqa: BLE001
            out[n] = {"kind": "error", "type": type(e).__name__}
            continue
        if isinstance(v, np.ndarray):
            out[n] = {"kind": "array", "shape": list(v.shape), "dtype": str(v.dtype)}
        elif callable(v):
            out[n] = {"kind": "method"}
        elif isinstance(v, (bool, int, float)):
            out[n] = {"kind": type(v).__name__, "value": v if np.isfinite(v) else str(v)}
        elif n == "__slots__":
            out[n] = {"kind": "slots", "value": list(v)}
        else:
            out[n] = {"kind": type(v).__name__}
    return out


def main():
    surface = {}
    for m in MODELS:
        opt = reference_factory(m)()
        surface[m] = {"optim": describe(opt), "params": describe(opt.params)}
    path = os.path.join(HERE, "genopt_surface.json")
    with open(path, "w") as fh:
        json.dump(surface, fh, indent=1, sort_keys=True)
    print(path, {m: len(v["optim"]) for m, v in surface.items()})


if __name__ == "__main__":
    main()
