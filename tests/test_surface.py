"""tpl_amd answers to every name tplcpp's bindings expose for the DP family.  The list of names is a fixture
(tests/golden/tplcpp_surface.json, written by tests/golden/make_surface_fixture.py from the reference's binding
sources: class / method / attribute / function names, nothing else).  No GPU: objects whose constructor creates a
device handle are checked on the class."""
import json
import os

import pytest

import tpl_amd

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "tplcpp_surface.json")))
# constructing these opens a device handle; their surface is checked on the class object
NEEDS_DEVICE = {"DynProgEnvironment", "DynProgLonPlanner", "PolyLatPlanner", "DynProgLatLonPlanner"}
# instance attributes set in __init__ of device-owning classes (def_readwrite members in the reference)
INSTANCE_ATTRS = {"DynProgEnvironment": {"ref_line"},
                  "DynProgLatLonPlanner": {"traj_dp", "traj_smooth", "traj_dp_cart", "traj_smooth_cart", "lat_dists"}}


def test_module_functions():
    for f in SURFACE["functions"]:
        assert callable(getattr(tpl_amd, f, None)), f"tpl_amd.{f} is missing"


@pytest.mark.parametrize("name", sorted(SURFACE["classes"]))
def test_class_surface(name):
    want = SURFACE["classes"][name]
    cls = getattr(tpl_amd, name, None)
    assert cls is not None, f"tpl_amd.{name} is missing"
    if name in NEEDS_DEVICE:
        obj, probe = None, cls
    else:
        obj = cls() if want["init"] or name == "RefLine" else None
        probe = obj if obj is not None else cls
    for m in want["methods"]:
        assert callable(getattr(probe, m, None)), f"{name}.{m}() is missing"
    for a in want["attributes"]:
        if name in NEEDS_DEVICE:
            assert a in INSTANCE_ATTRS.get(name, ()), f"{name}.{a} is not known to be set by __init__"
            continue
        assert hasattr(obj, a), f"{name}.{a} is missing"
        if name not in ("RefLine",):
            setattr(obj, a, getattr(obj, a))                    # read-write like def_readwrite
    if "slots" in want:
        assert list(obj.__slots__) == want["slots"], f"{name}.__slots__ differs"
    if "__deepcopy__" in want["methods"]:
        import copy
        other = copy.deepcopy(obj)
        assert other is not obj and other == obj


def test_device_classes_set_their_members(monkeypatch):
    """The def_readwrite members of the device-owning classes are assigned in __init__ (checked on the source of
    __init__, no handle is created)."""
    import inspect
    for name, attrs in INSTANCE_ATTRS.items():
        src = inspect.getsource(getattr(tpl_amd, name).__init__)
        for a in attrs:
            assert f"self.{a} =" in src, f"{name}.__init__ does not set {a}"


def test_ref_line_fromarray_and_lerp():
    """RefLine (utils.hpp:176-220): bounding-box centring, interpRefLine with its float weights, the short way
    round for the heading, the nearer sample's semantic flag (as the source writes it), clamping at both ends."""
    import numpy as np
    rl = np.zeros((4, 9))
    rl[:, 0] = [10.0, 12.0, 14.0, 16.0]
    rl[:, 1] = [-2.0, 0.0, 2.0, 8.0]
    rl[:, 2] = [3.1, -3.1, -3.0, -2.9]          # crosses +-pi between samples 0 and 1
    rl[:, 3] = [0.0, 2.0, 4.0, 6.0]
    rl[:, 5] = [5.0, 7.0, 9.0, 11.0]
    rl[:, 8] = [0.0, 1.0, 0.0, 1.0]
    line = tpl_amd.RefLine.fromarray(rl, 2.0)
    assert line.x_offset == 13.0 and line.y_offset == 3.0 and float(line.step_size) == 2.0
    pts = line.points
    assert len(pts) == 4 and pts[0].x == -3.0 and pts[3].y == 5.0 and pts[1].semantic == 1.0
    raw = tpl_amd.RefLine.fromarray(rl, 2.0, True)
    assert raw.x_offset == 0.0 and raw.points[0].x == 10.0
    p = raw.lerp(1.0)                              # halfway between samples 0 and 1
    assert p.x == 11.0 and p.y == -1.0 and p.s == 1.0 and p.v_max == 6.0
    want = 3.1 + (2 * np.pi - 6.2) * 0.5           # the short way round, not the mean (0.0)
    assert abs(p.heading - want) < 1e-12
    a = np.float32(0.5) / np.float32(2.0)          # s = 0.5: a = 0.25, ia = 0.75 -> ia < 0.5 is false -> sample i1
    q = raw.lerp(0.5)
    assert q.semantic == 1.0 and q.x == 10.0 * float(np.float32(1.0) - a) + 12.0 * float(a)
    assert raw.lerp(1.5).semantic == 0.0           # ia = 0.25 < 0.5 -> sample i0
    assert raw.lerp(-5.0).x == 10.0                # both indices clamp to sample 0: 10 * 3.5 + 10 * -2.5
    end = raw.lerp(100.0)                          # both indices clamp to the last sample; a = 50 - 3 = 47 extrapolates nothing
    assert end.x == 16.0 * (1.0 - 47.0) + 16.0 * 47.0
    with pytest.raises(ValueError, match="expected: \\(-1, 9\\)"):
        tpl_amd.RefLine.fromarray(np.zeros((3, 8)), 1.0)
    line.points = pts[:2]
    assert line.as_array().shape == (2, 9)


@pytest.mark.skipif(not os.path.isdir("/root/reference/library/tpl"), reason="the reference checkout exists in the build container only")
def test_the_references_own_util_module_runs_on_tpl_amd_as_tplcpp():
    """`import tpl_amd as tplcpp`: tpl/util.py (which imports nine names from tplcpp) imports and its wrappers work.
    numba / structstore / objtoolbox are absent from this image and stubbed; nothing of the reference is copied."""
    import subprocess
    import sys
    code = r'''
import sys, types
import numpy as np
import tpl_amd
sys.modules["tplcpp"] = tpl_amd
numba = types.ModuleType("numba"); numba.njit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
sys.modules["numba"] = numba
sys.modules["structstore"] = types.ModuleType("structstore")
otb = types.ModuleType("objtoolbox"); otb.bundle = lambda: types.SimpleNamespace(); sys.modules["objtoolbox"] = otb
sys.path.insert(0, "/root/reference/library")
import importlib.util
spec = importlib.util.spec_from_file_location("ref_util", "/root/reference/library/tpl/util.py")
tpl_pkg = types.ModuleType("tpl"); tpl_pkg.__path__ = ["/root/reference/library/tpl"]; sys.modules["tpl"] = tpl_pkg
optim = types.ModuleType("tpl.optim"); optim.__path__ = []; sys.modules["tpl.optim"] = optim
sys.modules["tpl.optim.optimizers"] = types.ModuleType("tpl.optim.optimizers"); optim.optimizers = sys.modules["tpl.optim.optimizers"]
util = importlib.util.module_from_spec(spec); spec.loader.exec_module(util)
path = np.stack([np.arange(40) * 0.5, np.zeros(40), np.zeros(40), np.arange(40) * 0.5, np.zeros(40), np.full(40, 5.0)], axis=1)
p = util.project(path[:, :2], [3.2, 1.0])
assert abs(p.arc_len - 3.2) < 1e-12 and abs(p.distance - 1.0) < 1e-12
rs = util.resample_path(path, 1.0, 10)
assert rs.shape == (10, 6) and abs(rs[3, 0] - 3.0) < 1e-12
assert abs(util.lerp(0.25, np.array([0.0, 1.0]), np.array([2.0, 4.0])) - 2.5) < 1e-12
assert util.lerp(np.array([0.25, 0.75]), np.array([0.0, 1.0]), np.array([[2.0, 0.0], [4.0, 1.0]])).shape == (2, 2)
assert util.point_in_polygon([0.5, 0.5], np.array([[0.0, 0], [1, 0], [1, 1], [0, 1]]))
print("ok")
'''
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def test_result_pool_never_hands_out_an_array_somebody_holds():
    """DynProgEnvironment's big getters recycle result arrays (tpl_amd/dyn_prog.py:_ResultPool): only arrays nobody
    refers to any more -- directly or through a view -- come back."""
    import numpy as np
    from tpl_amd.dyn_prog import _ResultPool
    pool = _ResultPool()
    f32 = np.dtype(np.float32)
    a = pool.take((4, 4), f32)
    a[:] = 7.0
    ida = id(a)
    b = pool.take((4, 4), f32)
    assert b is not a                                   # `a` is held by this test
    view = a[1:]
    del a
    c = pool.take((4, 4), f32)
    assert id(c) != ida and c is not b                  # a view keeps its base busy
    assert float(view[0, 0]) == 7.0
    del view
    d = pool.take((4, 4), f32)
    assert id(d) == ida                                 # released: recycled
    assert pool.take((4, 5), f32).shape == (4, 5)       # other shapes get their own arrays
    held = [pool.take((2, 2), f32) for _ in range(20)]  # more live results than the pool remembers: still all distinct
    assert len({id(x) for x in held}) == 20
