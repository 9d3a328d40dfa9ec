"""Planning-cycle dumps that pin the four DP-family engines against a binary of the reference, and their checker.

``tools/dump_cycle_goldens.py --module tplcpp`` (numpy + tplcpp + tests/golden/cycle_scenes.npz, nothing else) writes
one ``ref_cycle_<scene>.npz`` per scene on a CUDA box; a maintainer drops the files into tests/golden/.  None is
committed: the reference's engines need CUDA, 3-D texture hardware and OpenGL, which this image lacks ("parity
unpinned").  Arrays of a dump:

  occ_cartesian uint32[W, W]   get_occ_map_cartesian()            occ   float32[l, s, t]   get_occ_map()
  dist_lon float32[l, s, t, 2] get_dist_map_lon()                 lat_traj float64[s_steps, 12]  PolyLatTraj.as_numpy()
  lon_path float32[path_steps, 7]  the path handed to DynProgLonPlanner.update
  lon_traj float32[t_steps, 7] LonTraj.states rows (t, s, v, a, j, cost, constr);  lon_table_<t> optional
  ll_traj_dp float64[t_steps, F]  ll_table_<t> float32[l, ds, s, 4]  ll_traj_smooth  ll_traj_smooth_cart

Every stage is checked on the DUMPED output of the stage before it, so that one rasterisation difference does not
cascade: rasteriser + Frenet resampling + gap closing from the scene; distance map from the dumped occupancy; lattice
from the dumped occupancy; (s,v,a) plan from the dumped path and occupancy; (s,ds,l) plan from the dumped occupancy.
Bars: occupancy and distance maps identical; DP argmin actions identical, values 1e-5 relative; lattice trajectory
1e-5 relative (+1e-6 absolute); smoothed trajectory 1e-6.  Test infrastructure: nothing under tpl_amd/ imports it."""
import ctypes as C
import glob
import os

import numpy as np

import dp_golden_format as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "tests", "golden", "cycle_scenes.npz")
GOLDEN_GLOB = os.path.join(ROOT, "tests", "golden", "ref_cycle_*.npz")
NEED = ("scene", "occ_cartesian", "occ", "dist_lon", "lat_traj", "lon_path", "lon_traj", "ll_traj_dp", "ll_traj_smooth",
        "ll_traj_smooth_cart")


def golden_files():
    return sorted(glob.glob(GOLDEN_GLOB))


def fill(obj, names, vals):
    for n, v in zip(names, vals):
        n = str(n)
        cur = getattr(obj, n)
        setattr(obj, n, int(round(float(v))) if isinstance(cur, int) and not isinstance(cur, bool) else float(v))
    return obj


class Scene:
    """Inputs of one cycle, as parameter objects of tpl_amd (ctypes structs shared with the oracles)."""

    def __init__(self, name, scenes_path=SCENES):
        import tpl_amd
        z = np.load(scenes_path)
        if name not in [str(s) for s in z["scenes"]]:
            raise ValueError(f"scene {name!r} is not in {os.path.basename(scenes_path)}")
        g = lambda k: z[f"{name}.{k}"]
        self.name = name
        self.ep = fill(tpl_amd.DynProgEnvParams(), g("env_names"), g("env_vals"))
        self.ref_line, self.ref_step = np.ascontiguousarray(g("ref_line"), np.float64), float(g("ref_step"))
        self.geoms = [(np.ascontiguousarray(g(f"geom_{i}"), np.float64), bool(st)) for i, st in enumerate(g("geom_stationary"))]
        self.lp = fill(tpl_amd.PolyLatPlannerParams(), g("lat_names"), g("lat_vals"))
        self.lat_start = tpl_amd.PolyLatTrajPoint()
        s = self.lat_start
        s.s, s.l, s.dl, s.ddl, s.v = (float(v) for v in g("lat_start"))
        self.p = fill(tpl_amd.DynProgLonPlannerParams(), g("lon_names"), g("lon_vals"))
        self.lon_start = tpl_amd.LonState()
        self.lon_start.s, self.lon_start.v, self.lon_start.a = (float(v) for v in g("lon_start"))
        self.q = fill(tpl_amd.DynProgLatLonPlannerParams(), g("ll_names"), g("ll_vals"))
        self.ll_start = tpl_amd.LatLonState()
        self.ll_start.s, self.ll_start.ds, self.ll_start.l = (float(v) for v in g("ll_start"))


def load(path):
    z = np.load(path)
    missing = [k for k in NEED if k not in z.files]
    if missing:
        raise ValueError(f"{os.path.basename(path)}: missing arrays {missing}")
    d = {k: z[k] for k in z.files}
    sc = Scene(str(d["scene"]))
    ep, p, q = sc.ep, sc.p, sc.q
    shapes = {"occ": (ep.l_steps, ep.s_steps, ep.t_steps), "dist_lon": (ep.l_steps, ep.s_steps, ep.t_steps, 2),
              "lon_path": (p.path_steps, 7), "lon_traj": (p.t_steps, 7), "lat_traj": (sc.lp.s_steps, 12)}
    for k, want in shapes.items():
        if d[k].shape != want:
            raise ValueError(f"{os.path.basename(path)}: {k} has shape {d[k].shape}, the scene's parameters say {want}")
    if d["ll_traj_dp"].shape[0] != q.t_steps:
        raise ValueError(f"{os.path.basename(path)}: ll_traj_dp has {d['ll_traj_dp'].shape[0]} states, t_steps = {q.t_steps}")
    d["lon_tables"] = {int(k[10:]): d[k] for k in z.files if k.startswith("lon_table_")}
    d["ll_tables"] = {int(k[9:]): d[k] for k in z.files if k.startswith("ll_table_")}
    for t, tab in d["ll_tables"].items():
        if tab.shape != (q.l_steps, q.ds_steps, q.s_steps, 4):
            raise ValueError(f"{os.path.basename(path)}: ll_table_{t} has shape {tab.shape}")
    return sc, d


# --------------------------------------------------------------------------- the two implementations under test
def run_hip(sc, d):
    """Stage results of libtplhip.so, each stage fed with the dump's output of the stage before."""
    import tpl_amd
    out = {}
    env = tpl_amd.DynProgEnvironment()
    env.reinit_buffers(sc.ep)
    env.set_ref_line(sc.ref_line, sc.ref_step)
    for g, st in sc.geoms:
        env.insert_geometry(g, st)
    env.update()
    out["occ_cartesian"], out["occ"] = env.get_occ_map_cartesian(), env.get_occ_map()
    env.set_occ_map(d["occ"])
    env.update_dist_maps()
    out["dist_lon"] = env.get_dist_map_lon()
    lat = tpl_amd.PolyLatPlanner()
    lat.reinit_buffers(sc.lp)
    out["lat_traj"] = lat.update(sc.lat_start, env).as_numpy()
    dp = tpl_amd.DynProgLonPlanner()
    dp.reinit_buffers(sc.p)
    tr = dp.update(sc.lon_start, d["lon_path"], env)
    out["lon_traj"] = np.array([[s.t, s.s, s.v, s.a, s.j, s.cost, s.constr] for s in tr.states], dtype=np.float32)
    out["lon_tables"] = {t: dp.copy_table(t) for t in d["lon_tables"]}
    ll = tpl_amd.DynProgLatLonPlanner()
    ll.reinit_buffers(sc.q)
    ll.traj_dp[0] = sc.ll_start
    ll.traj_smooth[0] = sc.ll_start
    ll.update_traj_dp(env)
    out["ll_traj_dp"] = ll.traj_dp.as_numpy()
    out["ll_tables"] = {t: ll.copy_backward_pass_tex(t) for t in d["ll_tables"]}
    ll.update_traj_smooth(env)
    ll.update_traj_cart(env.ref_line)
    out["ll_traj_smooth"], out["ll_traj_smooth_cart"] = ll.traj_smooth.as_numpy(), ll.traj_smooth_cart.as_numpy()
    return out


def run_oracle(sc, d):
    """The same stages on the CPU restatements under oracle/."""
    import oracle_dp as O
    import oracle_env as oe
    import oracle_lat_lon as oll
    import oracle_poly_lat as opl
    from tpl_amd._capi import TplLatLonCartState, TplLatLonState, TplLonState
    out = {}
    o = oe.OracleEnv(sc.ep)
    o.set_ref_line(sc.ref_line, sc.ref_step)
    for g, st in sc.geoms:
        o.insert_geometry(g, st)
    out["occ_cartesian"], out["occ"], _ = o.update()
    occ = np.ascontiguousarray(d["occ"], np.float32)
    dl = np.zeros(occ.shape + (2,), np.float32)
    oe.lib().dpo_env_dist_map_lon(C.byref(o.c), occ.ctypes.data_as(C.POINTER(C.c_float)), dl.ctypes.data_as(C.POINTER(C.c_float)))
    out["dist_lon"] = dl
    _, _, pts, _ = opl.plan(sc.lp, o.c, sc.lat_start, o.ref, o.step, o.offsets, occ)
    out["lat_traj"] = pts
    # (s,v,a) plan
    p, ep = sc.p._c, o.c
    lib = O.lib()
    pc = type(p)()
    C.memmove(C.byref(pc), C.byref(p), C.sizeof(p))
    lib.dpo_update_step_sizes(C.byref(pc))
    epc = type(ep)()
    C.memmove(C.byref(epc), C.byref(ep), C.sizeof(ep))
    lib.dpo_env_update_step_sizes(C.byref(epc))
    path = np.ascontiguousarray(d["lon_path"], np.float32)
    dist = O.dist_map_path(epc, occ, path, pc.path_steps, pc.path_step_size)
    cin = (TplLonState * 1)()
    C.memmove(C.byref(cin[0]), C.byref(sc.lon_start._c), C.sizeof(TplLonState))
    tables = np.zeros((pc.t_steps, pc.a_steps, pc.v_steps, pc.s_steps, 4), dtype=np.float32)
    res = (TplLonState * pc.t_steps)()
    assert lib.dpo_update(C.byref(pc), C.byref(epc), O.fptr(path), O.fptr(dist), 1, cin, O.fptr(tables), res, 0) == 0
    out["lon_traj"] = O.states_to_np(res).reshape(pc.t_steps, 7)
    out["lon_tables"] = {t: tables[t] for t in d["lon_tables"]}
    # (s,ds,l) plan
    ll_tables, otraj = oll.update_traj_dp(sc.q, o.c, dl, o.ref, o.step, sc.ll_start)
    out["ll_traj_dp"] = oll.states_np(otraj)[:, :9]                       # LatLonTraj.as_numpy(): t .. dddl
    out["ll_tables"] = {t: ll_tables[t] for t in d["ll_tables"]}
    sm = (TplLatLonState * 512)()
    n = oll.lib().llo_update_traj_smooth(C.byref(sc.q._c), otraj, len(otraj), C.byref(sc.ll_start._c), sm, 512)
    out["ll_traj_smooth"] = oll.states_np(sm[:n])[:, :9]
    cart = (TplLatLonCartState * n)()
    oll.lib().llo_frenet_to_cartesian(o.ref.ctypes.data_as(C.POINTER(C.c_double)), len(o.ref), o.step,
                                      o.offsets.ctypes.data_as(C.POINTER(C.c_double)), sm, n, cart)
    out["ll_traj_smooth_cart"] = oll.states_np(cart, fields=oll.CART_F)[:, :8]       # LatLonCartTraj.as_numpy(): t .. k
    return out


# --------------------------------------------------------------------------- the bars
def check(sc, d, got, who):
    """Raises AssertionError naming the first stage outside its bar; returns the per-stage figures."""
    rep = {}
    for k in ("occ_cartesian", "occ", "dist_lon"):
        a, b = np.asarray(got[k]), np.asarray(d[k])
        assert a.shape == b.shape, f"{who}: {k} has shape {a.shape}, the dump {b.shape}"
        n = int(np.count_nonzero(a != b))
        rep[k] = n
        assert n == 0, f"{who}: {k}: {n} of {a.size} cells differ from the reference dump"
    a, b = got["lat_traj"], d["lat_traj"]
    assert a.shape == b.shape and np.allclose(a, b, rtol=1e-5, atol=1e-6), \
        f"{who}: lateral trajectory differs by {np.max(np.abs(a - b)):.3e}"
    g = dict(p=sc.p._c, trajs=d["lon_traj"][None], tables=d["lon_tables"])
    G.check_against(g, np.asarray(got["lon_traj"], np.float32)[None], got["lon_tables"], who=who)
    a, b = got["ll_traj_dp"], d["ll_traj_dp"]
    assert a.shape == b.shape and np.allclose(a, b, rtol=1e-5, atol=1e-6), f"{who}: (s,ds,l) DP trajectory differs"
    for t, ref in d["ll_tables"].items():
        tab = got["ll_tables"][t]
        # channel 2/3 of a cell hold the chosen action (lat_lon_planner.cu); all four channels are point-sampled floats
        n = int(np.count_nonzero(tab.view(np.uint32) != ref.view(np.uint32)))
        rep[f"ll_table_{t}"] = n
        assert np.allclose(tab, ref, rtol=1e-5, atol=1e-6), f"{who}: (s,ds,l) stage {t}: values differ, {n} not bit-identical"
    for k in ("ll_traj_smooth", "ll_traj_smooth_cart"):
        a, b = got[k], d[k]
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-6, atol=1e-6), f"{who}: {k} differs"
    return rep
