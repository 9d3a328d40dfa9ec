"""The C-ABI library loads without a GPU and exports every symbol the headers declare."""
import ctypes as C
import glob
import os
import re

import pytest

from tpl_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="tplhip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    return re.findall(r"TPLHIP_API\s+[\w\s\*]+?\b(tplhip_\w+)\s*\(", text)


def test_every_declared_symbol_is_exported():
    lib = C.CDLL(_capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported by libtplhip.so"


def test_python_signatures_cover_the_header():
    assert set(_declared()) == set(_capi.SIGNATURES)
    from tpl_amd.optim import solver
    assert set(_declared("tplhip_ilqr.h")) == set(solver.SIGNATURES)


def test_every_model_library_exports_the_ilqr_abi():
    from tpl_amd.optim import optimizers as o, solver
    names = _declared("tplhip_ilqr.h")
    assert len(names) >= 20
    for fn in o.CONFIG_FUNCTIONS + o.EXTRA_CONFIG_FUNCTIONS:
        cfg = fn()
        so = os.path.join(ROOT, "tpl_amd", "models", f"libilqr_{cfg.name}.so")
        lib = solver.load_model(so)                 # declares every signature or raises
        for n in names:
            assert hasattr(lib, n), (cfg.name, n)
        x, u, c, npar = (C.c_int(), C.c_int(), C.c_int(), C.c_int())
        lib.tplhip_ilqr_dims(C.byref(x), C.byref(u), C.byref(c), C.byref(npar))
        assert (x.value, u.value, c.value) == (len(cfg.states), len(cfg.actions), len(cfg.constraints))
        assert [lib.tplhip_ilqr_param_name(i).decode() for i in range(npar.value)] == [p.name for p in cfg.param_symbols]
        assert lib.tplhip_ilqr_fingerprint().decode() == cfg.fingerprint()


def test_no_gpu_means_loud_failure():
    lib = _capi.load()
    if lib.tplhip_device_count() > 0:
        pytest.skip("a GPU is present")
    import tpl_amd
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        tpl_amd.DynProgLonPlanner()
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        tpl_amd.DynProgEnvironment()
    from tpl_amd.optim import genopt, optimizers as o
    make = genopt.build(o.config_simple_vehicle(), "simple_vehicle")
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        make()


def test_param_surface_matches_reference_slots():
    import tpl_amd
    p = tpl_amd.DynProgLonPlannerParams()
    # bindings_dp_lon.cu:36-64 order
    assert p.__slots__ == [
        "s_min", "s_max", "v_min", "v_max", "a_min", "a_max", "j_min", "j_max",
        "t_steps", "s_steps", "v_steps", "a_steps", "dt", "dt_start", "time_gap", "gap_min",
        "w_progress", "w_a", "w_j", "w_snap", "w_safety_dist", "path_step_size", "path_steps",
        "width_veh", "length_veh"]
    # lon_planner.cuh:9-43 defaults
    assert (p.s_steps, p.v_steps, p.a_steps, p.t_steps) == (201, 37, 7, 10)
    assert p.w_safety_dist == 10.0 and p.length_veh == 6.0 and p.path_step_size == 0.5
    e = tpl_amd.DynProgEnvParams()
    assert e.__slots__[-4:] == ["t_steps", "l_steps", "s_steps", "dir_steps"]
    assert (e.t_steps, e.l_steps, e.s_steps) == (10, 21, 201)
    s = tpl_amd.LonState()
    s.s, s.v, s.a, s.j = 1, 2, 3, 4
    assert list(s.as_numpy()) == [1.0, 2.0, 3.0, 4.0]
    with pytest.raises(AttributeError):
        p.nonexistent = 1


def test_every_selectable_rule_has_a_built_sweep():
    """Every (arithmetic rule, division forms) pair the host can select for a plain-fp32 sweep has a built kernel instance
    (tplhip_dp_lon_sweep_built: host only, launches nothing).  pick_div_forms (dp_lon.hip) gives the texture-filter defaults
    (R2 rint, R3 fma form, a -> v -> s) -- under BOTH settings of R1 -- the cheapest proven form per divisor, i.e. any pair of
    {4, 3, 5, 0}; every other rule both divisions in the Newton form (5, 5) or both IEEE (0, 0); another blend order runs
    k_backward3o for the weight rules rint and floor (un-quantised weights take the general sweep: use_v3) and has no
    one-cell-per-lane form.  A selectable pair without an instance would send its stages to the general sweep silently
    (ADVICE r5): this is the test that says the two sides agree."""
    lib = _capi.load()
    built = lib.tplhip_dp_lon_sweep_built
    forms = (4, 3, 5, 0)
    n = 0
    for quant in (0, 1, 2):
        for lerp in (0, 1, 2):
            for fma in (0, 1):
                for order in (0, 1, 2, 3):
                    all_forms = (quant, lerp, order) == (1, 0, 0)
                    pairs = [(a, b) for a in forms for b in forms] if all_forms else [(5, 5), (0, 0)]
                    for pd, rd in pairs:
                        if order == 0:
                            assert built(quant, lerp, fma, order, pd, rd, 1) == 1, (quant, lerp, fma, order, pd, rd, "k_backward3v")
                            assert built(quant, lerp, fma, order, pd, rd, 2) == 1, (quant, lerp, fma, order, pd, rd, "k_backward3s")
                            n += 2
                        elif quant in (1, 2):
                            assert built(quant, lerp, fma, order, pd, rd, 1) == 1, (quant, lerp, fma, order, pd, rd, "k_backward3o")
                            n += 1
                        else:
                            assert built(quant, lerp, fma, order, pd, rd, 1) == 0      # not built, and use_v3 never selects it
    assert n == 2 * (2 * 16 + 16 * 2) + 3 * 2 * 3 * 2 * 2
    # the chained form of k_backward3v (kind 3): every pair of forms under the default filter rule and order, either R1 -- what
    # build_action_tables (dp_lon.hip) makes eligible -- and nothing else
    for quant in (0, 1, 2):
        for lerp in (0, 1, 2):
            for fma in (0, 1):
                for pd in forms:
                    for rd in forms:
                        assert built(quant, lerp, fma, 0, pd, rd, 3) == (1 if (quant, lerp) == (1, 0) else 0), (quant, lerp, fma, pd, rd)
    assert built(1, 0, 1, 1, 5, 5, 3) == 0
    # what is NOT selectable is not claimed: a non-default filter rule with a mixed pair of forms
    assert built(2, 0, 1, 0, 4, 3, 1) == 0 and built(1, 0, 1, 0, 7, 7, 1) == 0


def test_switches_refuse_a_null_handle():
    """The run-time switches of the planner (filter rule, contraction, blend order, chained sweep) check their handle before
    touching a device: callable without a GPU."""
    lib = _capi.load()
    for call in (lambda: lib.tplhip_dp_lon_set_chained_sweep(None, 1), lambda: lib.tplhip_dp_lon_set_filter_rule(None, 1, 0),
                 lambda: lib.tplhip_dp_lon_set_contraction(None, 1), lambda: lib.tplhip_dp_lon_set_blend_order(None, 0)):
        assert call() == _capi.TPLHIP_EINVAL
        assert "null handle" in _capi.last_error()
