"""The generated solver object answers to every name of the reference's generated type, with the same array shapes,
dtypes, scalar types and default values.  The names come from the reference module itself, imported in the build
container (tests/golden/genopt_surface.json, written by tests/golden/make_genopt_surface.py)."""
import json
import os

import numpy as np
import pytest

from tpl_amd.optim import genopt, optimizers as o

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "genopt_surface.json")))


def _compare(obj, want, where):
    for name, w in want.items():
        assert hasattr(obj, name), f"{where}.{name} is missing"
        v = getattr(obj, name)
        if w["kind"] == "array":
            assert isinstance(v, np.ndarray), f"{where}.{name} is not an array"
            assert list(v.shape) == w["shape"] and str(v.dtype) == w["dtype"], \
                f"{where}.{name}: {v.shape} {v.dtype}, the reference has {w['shape']} {w['dtype']}"
        elif w["kind"] == "method":
            assert callable(v), f"{where}.{name} is not callable"
        elif w["kind"] == "slots":
            assert list(v) == w["value"], f"{where}.__slots__ differs: {list(v)} vs {w['value']}"
        elif w["kind"] in ("int", "float", "bool"):
            assert type(v).__name__ == w["kind"] or (w["kind"] == "float" and isinstance(v, float)), \
                f"{where}.{name} is {type(v).__name__}, the reference's is {w['kind']}"
            if name not in ("runtime", "elapsed_update_time"):
                assert v == w["value"], f"{where}.{name} defaults to {v!r}, the reference's to {w['value']!r}"


@pytest.mark.parametrize("model", sorted(SURFACE))
def test_object_surface(model):
    make = genopt.build(getattr(o, "config_" + model)(), model)
    opt = make()
    _compare(opt, SURFACE[model]["optim"], model)
    _compare(opt.params, SURFACE[model]["params"], model + ".params")
