"""Known-answer tests of the occupancy-pipeline oracle (oracle/dp_env_oracle.c): hand-derivable
cases of the rules its header freezes.  No GPU."""
import numpy as np
import pytest

import oracle_env as oe
from benchkit import scenarios as sc


def _env(t_steps=4, s_steps=201, l_steps=31, **kw):
    ep = sc.env_params(t_steps=t_steps, s_steps=s_steps, l_steps=l_steps)
    for k, v in kw.items():
        setattr(ep, k, v)
    return ep


def _axis_line(n=440, step=0.5, semantic_range=None):
    # along +x from the origin: after centring, x in [-109.75, 109.75], y = 0
    return sc.ref_line_straight(n=n, step=step, x0=0.0, y0=0.0, heading=0.0, semantic_range=semantic_range)


def _quad(x0, x1, y0, y1, t, split="a"):
    p = [(x0, y0), (x1, y0), (x0, y1), (x0, y1), (x1, y0), (x1, y1)] if split == "a" else \
        [(x0, y0), (x1, y0), (x1, y1), (x0, y0), (x1, y1), (x0, y1)]
    return np.array([(a, b, t) for a, b in p], dtype=np.float64)


def test_centering_and_image_size():
    ep = _env()
    o = oe.OracleEnv(ep)
    rl = _axis_line()
    o.set_ref_line(rl, 0.5)
    assert o.offsets[0] == pytest.approx(109.75) and o.offsets[1] == 0.0          # bounding-box centre
    assert oe.lib().dpo_env_cart_size(oe.C.byref(o.c)) == 840                     # ceil((200 + 2*5) * 4)


def test_pixel_aligned_square_and_shared_edge():
    """A square on pixel boundaries covers exactly its pixels; the two triangles of the quad share
    the diagonal without double cover or gap (top-left rule), whichever way the quad is split."""
    ep = _env()
    counts = []
    for split in ("a", "b"):
        o = oe.OracleEnv(ep)
        o.set_ref_line(_axis_line(), 0.5)
        g = _quad(109.75 + 10.0, 109.75 + 12.0, -1.0, 1.0, 0.0, split)       # 8 x 8 pixels
        g[3:, 2] = 1.0                                                        # second triangle: slice 1
        o.insert_geometry(g, False)
        img, _, _ = o.update()
        assert np.count_nonzero(img) == 64
        assert set(np.unique(img)) == {0, 1, 2}                               # never both bits
        ys, xs = np.nonzero(img)
        assert xs.min() == 420 + 40 and xs.max() == 420 + 47 and ys.min() == 416 and ys.max() == 423
        counts.append((int((img == 1).sum()), int((img == 2).sum())))
    assert counts[0][0] + counts[0][1] == 64 and counts[1][0] + counts[1][1] == 64


def test_both_windings_and_degenerate_triangles():
    ep = _env()
    o = oe.OracleEnv(ep)
    o.set_ref_line(_axis_line(), 0.5)
    g = _quad(120.0, 124.0, -1.0, 1.0, 0.0)
    o.insert_geometry(g, False)
    img_ccw, _, _ = o.update()
    o.insert_geometry(g[[0, 2, 1, 3, 5, 4]], False)                           # reversed winding
    img_cw, _, _ = o.update()
    assert np.array_equal(img_ccw, img_cw) and np.count_nonzero(img_ccw) > 0
    line = np.array([(120.0, 0.0, 0.0), (122.0, 0.0, 0.0), (124.0, 0.0, 0.0)])
    o.insert_geometry(line, False)
    img, _, _ = o.update()
    assert np.count_nonzero(img) == 0
    o.insert_geometry(g[:5], False)                                           # partial triangle dropped
    img, _, _ = o.update()
    assert np.count_nonzero(img) == np.count_nonzero(img_ccw) // 2


def test_time_flags_and_frenet_values():
    """Slice flags, stationary objects, road bounds, border cells (env.cu:45-60)."""
    ep = _env(t_steps=4)
    o = oe.OracleEnv(ep)
    rl = _axis_line()
    rl[:, 6] = 3.0          # d_left
    rl[:, 7] = 2.0          # d_right
    o.set_ref_line(rl, 0.5)
    o.insert_geometry(_quad(50.0, 54.0, -1.0, 1.0, 3.0), False)               # t = 3 s -> slice 3 (env.cu:363)
    o.insert_geometry(_quad(100.0, 104.0, -1.0, 1.0, 0.0), True)              # stationary
    img, occ, _ = o.update()
    assert set(np.unique(img)) == {0, 1 << 3, (1 << 0) | (1 << 31)}
    s_idx = lambda s: int(round(s / 1.0))
    l_idx = lambda l: int(round((l + 5.0) / (10.0 / 30.0)))
    assert occ[l_idx(0.0), s_idx(52.0), 3] == 1.0 and occ[l_idx(0.0), s_idx(52.0), 2] == 0.0
    assert np.all(occ[l_idx(0.0), s_idx(102.0), :] == 3.0)                    # stationary in every slice
    assert np.all(occ[0] == 2.0) and np.all(occ[-1] == 2.0)                   # border cells
    assert np.all(occ[l_idx(3.4)] == 2.0) and np.all(occ[l_idx(-2.4)] == 2.0)  # beyond d_left / d_right
    assert occ[l_idx(0.0), s_idx(20.0), 0] == 0.0
    # dilation: ceil(1.414 * 4) = 6 pixels = 1.5 m -> a cell 1.3 m beside the box is hit, 2.0 m is not
    assert occ[l_idx(0.0), s_idx(48.9), 3] == 1.0 or occ[l_idx(0.0), s_idx(49.0), 3] == 1.0
    assert occ[l_idx(0.0), s_idx(47.0), 3] == 0.0


def test_speed_limit_zero_closes_the_road():
    ep = _env()
    o = oe.OracleEnv(ep)
    rl = _axis_line()
    rl[200:, 5] = 0.0
    o.set_ref_line(rl, 0.5)
    _, occ, _ = o.update()
    assert np.all(occ[:, 101:, :] == 2.0) and np.all(occ[2:-2, :99, :] == 0.0)      # |l| <= d_left = d_right = 4.5


def test_close_intersections_and_distance_maps():
    ep = _env(t_steps=3)
    o = oe.OracleEnv(ep)
    rl = _axis_line(semantic_range=(100, 140))                                # stations 50 .. 70 m
    o.set_ref_line(rl, 0.5)
    o.insert_geometry(_quad(60.0, 62.0, 2.0, 3.0, 2.0), False)                # slice 2, left part of the lane
    o.insert_geometry(_quad(120.0, 122.0, 2.0, 3.0, 2.0), False)              # outside the intersection
    _, occ, dl = o.update()
    l_mid = 15
    assert occ[l_mid, 61, 2] == 1.0          # closed across the whole lane inside the intersection
    assert occ[1, 61, 2] == 1.0 and occ[29, 61, 2] == 1.0 and occ[0, 61, 2] == 2.0
    assert occ[l_mid, 121, 2] == 0.0         # not outside it
    assert occ[l_mid, 61, 1] == 0.0
    # distance ahead / behind along s in the middle lane cell, slice 2
    first = int(np.argmax(occ[l_mid, :, 2] > 0))
    assert dl[l_mid, first, 2, 0] == 0.0
    assert dl[l_mid, first - 3, 2, 0] == pytest.approx(3.0)
    last = 200 - int(np.argmax(occ[l_mid, ::-1, 2] > 0))
    assert dl[l_mid, last + 4, 2, 1] == pytest.approx(4.0)
    assert dl[l_mid, 0, 1, 0] == pytest.approx(10000.0 + 201.0)               # free lane: 10000 + s_steps * step
    assert dl[0, 5, 0, 0] == 0.0                                              # border cells count as occupied


def test_short_reference_line_is_refused():
    ep = _env()
    o = oe.OracleEnv(ep)
    o.set_ref_line(_axis_line(n=300), 0.5)                                    # 150 m < s_max = 200 m
    assert o.update() is None
