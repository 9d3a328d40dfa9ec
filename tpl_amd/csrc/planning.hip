// planning.hip -- host-side geometry and profile helpers of the path/velocity-decomposition glue
// (SURVEY 8 f-4).  Nothing here runs on the GPU: these are the small serial routines the reference keeps in its
// C++ extension (tplcpp.project / resample, library/src/utils.cpp:257-548) or compiles with numba
// (rampify_profile, library/tpl/planning/utils.py:5-65 and planning/path_vel_decomp/path_optim.py:11-59), which
// the Python callers of the iLQR solvers run every planning cycle.  Restated from those sources; the order of the
// floating-point operations follows them (the numba versions are fastmath, so they are only defined up to
// reassociation; tests compare at 1e-9).
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <vector>

#include "tplhip_common.h"
#include "dp_env_internal.h"      // short_angle_dist (utils.hpp:21-53)

namespace {

inline long mod_floor(long a, long n) { long r = a % n; return r < 0 ? r + n : r; }

// utils.cpp:257-408
void project_one(const double* pts, long stride, long n, bool closed, double px, double py, TplProjection* out)
{
    TplProjection proj = {};
    double best = INFINITY, off = 0.0, arc_len = 0.0;
    const long end = closed ? n + 1 : n;
    double prev_x = pts[0], prev_y = pts[1];
    for (long i = 1; i < end; ++i) {
        const long ii = mod_floor(i, n);
        const double nx = pts[ii * stride], ny = pts[ii * stride + 1];
        const double pvx = px - prev_x, pvy = py - prev_y;
        const double vx = nx - prev_x, vy = ny - prev_y;
        const double l = std::sqrt(vx * vx + vy * vy);
        double q = (pvx * vx + pvy * vy) / (vx * vx + vy * vy);
        double cx, cy;
        bool in_bounds = true;
        if (q < 0) { in_bounds = !closed && i != 1; q = 0.0; cx = prev_x; cy = prev_y; }
        else if (q > 1) { in_bounds = !closed && i != end - 1; q = 1.0; cx = nx; cy = ny; }
        else { cx = prev_x + vx * q; cy = prev_y + vy * q; }
        const double dx = px - cx, dy = py - cy;
        const double d = dx * dx + dy * dy;
        if (d < best) {
            proj.in_bounds = in_bounds ? 1 : 0;
            best = d;
            proj.point[0] = cx; proj.point[1] = cy;
            proj.index = i;
            proj.alpha = q;
            off = arc_len;
        }
        arc_len += l;
        prev_x = nx; prev_y = ny;
    }
    proj.distance = std::sqrt(best);
    long idx_start, idx_end, idx_next;
    if (closed) {
        idx_start = mod_floor(proj.index - 1, n);
        idx_end = mod_floor(proj.index, n);
        idx_next = mod_floor(proj.index + 1, n);
    } else {
        idx_start = std::max(0L, (long)proj.index - 1);
        idx_end = proj.index;
        idx_next = std::min(n - 1, (long)proj.index + 1);
    }
    proj.start = idx_start;
    proj.end = idx_end;
    if (proj.alpha < 0.5) proj.index = idx_start;
    const double ax = pts[idx_start * stride], ay = pts[idx_start * stride + 1];
    const double bx = pts[idx_end * stride], by = pts[idx_end * stride + 1];
    double vx = bx - ax, vy = by - ay;
    const double l = std::sqrt(vx * vx + vy * vy);
    vx /= l; vy /= l;
    const double adx = ax - proj.point[0], ady = ay - proj.point[1];
    proj.arc_len = off + std::sqrt(adx * adx + ady * ady) * ((proj.alpha < 0) ? -1.0 : 1.0);
    if (proj.index < n - 2) {
        double wx = pts[idx_next * stride] - bx, wy = pts[idx_next * stride + 1] - by;
        const double wl = std::sqrt(wx * wx + wy * wy);
        wx /= wl; wy /= wl;
        proj.tangent[0] = proj.alpha * wx + (1.0 - proj.alpha) * vx;
        proj.tangent[1] = proj.alpha * wy + (1.0 - proj.alpha) * vy;
    } else {
        proj.tangent[0] = vx; proj.tangent[1] = vy;
    }
    proj.angle = std::atan2(proj.tangent[1], proj.tangent[0]);
    double ox = proj.point[0] - px, oy = proj.point[1] - py;
    const double ol = std::sqrt(ox * ox + oy * oy);
    ox /= ol; oy /= ol;                                   // a point ON the line gives NaN here, like the reference
    const double rx = -oy, ry = ox;
    if (vx * rx + vy * ry <= 0) proj.distance *= -1.0;
    *out = proj;
}

// InterpVars(vals, &PredictionPoint::t, v), utils.hpp:58-86, on rows of 5 doubles (t first)
struct PredInterp { long i_prev, i_next; double a; };
PredInterp pred_interp(const double* pred, long n, double v)
{
    PredInterp r = {0, 0, 0.0};
    if (v < pred[0]) { r.i_prev = 0; r.i_next = 1; }
    else if (v >= pred[(n - 1) * 5]) { r.i_prev = n - 2; r.i_next = n - 1; }
    else {
        for (long i = 1; i < n; ++i) {
            if (pred[(i - 1) * 5] <= v && v < pred[i * 5]) { r.i_prev = i - 1; r.i_next = i; break; }
        }
    }
    const double t_prev = pred[r.i_prev * 5], t_next = pred[r.i_next * 5];
    r.a = std::max(0.0, std::min(1.0, (v - t_prev) / (t_next - t_prev)));
    return r;
}

// interpPrediction, utils.cpp:560-574: out = (t, x, y, heading, v)
void interp_prediction(const double* pred, long n, double t, double* out)
{
    const PredInterp i = pred_interp(pred, n, t);
    const double* p = pred + i.i_prev * 5;
    const double* q = pred + i.i_next * 5;
    out[0] = t;
    out[1] = p[1] * (1.0 - i.a) + q[1] * i.a;
    out[2] = p[2] * (1.0 - i.a) + q[2] * i.a;
    out[3] = p[3] + tplhip::short_angle_dist(p[3], q[3]) * i.a;
    out[4] = p[4] * (1.0 - i.a) + q[4] * i.a;
}

} // namespace

extern "C" {

int tplhip_path_project(const double* points, int stride, int n, int closed, const double* positions, int m,
                        TplProjection* out)
{
    if (!points || !positions || !out || stride < 2) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (n < 2) return tplhip_fail(TPLHIP_EINVAL, "a line strip needs at least two points (got %d)", n);
    for (int k = 0; k < m; ++k) project_one(points, stride, n, closed != 0, positions[2 * k], positions[2 * k + 1], out + k);
    return TPLHIP_OK;
}

// utils.cpp:410-548: out = double[steps][5] = (x, y, alpha, prev index, next index); *n_out rows are valid
int tplhip_path_resample(const double* pts_in, int stride, int n, double sampling_dist, int steps, int start_index,
                         int closed_i, double* out, int* n_out)
{
    if (!pts_in || !out || !n_out || stride < 2) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    const bool closed = closed_i != 0;
    *n_out = 0;
    if (n == 0 || steps == 0) return TPLHIP_OK;
    std::vector<double> px(n), py(n);
    px[0] = pts_in[0]; py[0] = pts_in[1];
    long count_pts = 1;
    for (long k = 1; k < n; ++k) {                       // drops duplicate points
        px[count_pts] = pts_in[k * stride]; py[count_pts] = pts_in[k * stride + 1];
        const double dx = px[count_pts] - px[count_pts - 1], dy = py[count_pts] - py[count_pts - 1];
        if (std::sqrt(dx * dx + dy * dy) != 0) count_pts += 1;
    }
    if (count_pts == 1) {
        out[0] = px[0]; out[1] = py[0]; out[2] = out[3] = out[4] = 0.0;
        *n_out = 1;
        return TPLHIP_OK;
    }
    long start = closed ? mod_floor(start_index, count_pts) : std::max(0L, std::min((long)start_index, count_pts - 1));
    for (long i = 0; i < (long)steps * 5; ++i) out[i] = 0.0;
    out[0] = px[start]; out[1] = py[start]; out[2] = 0.0; out[3] = (double)start;
    out[4] = closed ? (double)mod_floor(start + 1, count_pts) : (double)std::max(0L, std::min(start + 1, count_pts - 1));
    long count = 1, i = start;
    while (count < steps) {
        const long prev_count = count;
        for (long k = 0; k < count_pts; ++k) {
            long prev_idx = i + k, next_idx = i + k + 1;
            if (closed) { prev_idx = mod_floor(prev_idx, count_pts); next_idx = mod_floor(next_idx, count_pts); }
            else { prev_idx = std::max(0L, std::min(prev_idx, count_pts - 2)); next_idx = std::max(0L, std::min(next_idx, count_pts - 1)); }
            const double ax = px[prev_idx], ay = py[prev_idx], bx = px[next_idx], by = py[next_idx];
            const double vx = bx - ax, vy = by - ay;
            const double l = std::sqrt(vx * vx + vy * vy), ls = l * l;
            const double ux = vx / l, uy = vy / l;
            const double cx = out[(count - 1) * 5], cy = out[(count - 1) * 5 + 1];
            // intersection of the segment with the circle of radius sampling_dist around the last sample
            const double D = (ax - cx) * (by - cy) - (bx - cx) * (ay - cy);
            const double disc = sampling_dist * sampling_dist * ls - D * D;
            if (disc < 0) return tplhip_fail(TPLHIP_ESTATE, "cannot solve for next sampling point");
            const double sq = std::sqrt(disc);
            const double sign_y = (vy < 0.0) ? -1.0 : 1.0;
            const double xp0 = D * vy, yp0 = -D * vx, xp1 = sign_y * vx * sq, yp1 = std::fabs(vy) * sq;
            double p0x = (xp0 + xp1) / ls + cx, p0y = (yp0 + yp1) / ls + cy;
            const double p1x = (xp0 - xp1) / ls + cx, p1y = (yp0 - yp1) / ls + cy;
            double q0 = (ux * (p0x - ax) + uy * (p0y - ay)) / l;
            const double q1 = (ux * (p1x - ax) + uy * (p1y - ay)) / l;
            const double tol = 1e-8;
            if (q0 < q1) { q0 = q1; p0x = p1x; p0y = p1y; }
            if ((!closed && next_idx == count_pts - 1) || (q0 > -tol && q0 - 1.0 < tol)) {
                i = prev_idx;
                double* s = out + count * 5;
                s[0] = p0x; s[1] = p0y; s[2] = q0; s[3] = (double)prev_idx; s[4] = (double)next_idx;
                count += 1;
                break;
            }
        }
        if (count == prev_count) return tplhip_fail(TPLHIP_ESTATE, "resampling failed");
    }
    *n_out = (int)count;
    return TPLHIP_OK;
}

// planning/utils.py:5-65.  out = double[n][2] = (v, a) per station.
int tplhip_rampify_velocity_profile(int has_v0, double v0, int has_a0, double a0, const double* lim_v_in, int n,
                                    double a_min, double a_max, double j_min, double j_max, double v_min, double step,
                                    double* out)
{
    if (!lim_v_in || !out || n < 1) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    std::vector<double> lim_v(n);
    for (int i = 0; i < n; ++i) lim_v[i] = std::max(lim_v_in[i], v_min);
    for (int i = 0; i < 2 * n; ++i) out[i] = 0.0;
    // backward pass: minimum jerk and acceleration
    double cv = lim_v[n - 1], ca = 0.0;
    for (int t = n - 1; t > 0; --t) {
        out[2 * t] = cv; out[2 * t + 1] = ca;
        const double lim_a = std::max(a_min, (cv - lim_v[t - 1]) / step * cv);
        if (lim_a < 0.0) ca = std::max(ca + j_min / cv * step, lim_a);
        else { ca = 0.0; cv = lim_v[t]; }
        cv += std::min(-ca / cv * step, lim_v[t - 1] - cv);
    }
    // forward pass: maximum jerk and acceleration
    if (!has_v0) out[0] = cv;
    else { cv = std::max(v0, v_min); out[0] = std::max(v0, v_min); }
    if (!has_a0) { ca = -ca; out[1] = ca; }
    else { ca = a0; out[1] = a0; }
    double lim_a = 0.0;
    for (int t = 0; t < n; ++t) {
        if (t < n - 1) lim_a = std::min(a_max, (out[2 * (t + 1)] - cv) / step * cv);
        if (lim_a > 0.0) ca = std::min(ca + j_max / cv * step, lim_a);
        else { ca = 0.0; cv = out[2 * t]; }
        const double next_v = cv + std::min(ca / cv * step, lim_v[t] - cv);
        cv = std::min(out[2 * t], next_v);
        out[2 * t] = cv; out[2 * t + 1] = ca;
    }
    return TPLHIP_OK;
}

// planning/path_vel_decomp/path_optim.py:11-59.  path_v = the speed-limit column of the path (path[:, 5]) with its
// stride in doubles; lower / upper: double[horizon]; out: double[n_path].
int tplhip_rampify_lateral_profile(double step, int horizon, double evasion_sharpness, double proj_distance,
                                   const double* path_v, int path_stride, int n_path, double gap,
                                   const double* lower, const double* upper, double* out)
{
    if (!path_v || !lower || !upper || !out || horizon < 1 || n_path < horizon)
        return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    std::vector<double> fwd(n_path, -10.0), bwd(n_path, -10.0);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<double>& pd = pass == 0 ? fwd : bwd;
        double d = pass == 0 ? lower[0] : lower[horizon - 1];
        for (int c = 0; c < horizon; ++c) {
            const int i = pass == 0 ? c : horizon - 1 - c;
            d = std::max(lower[i], d);
            pd[i] = d;
            const double v = std::max(path_v[(long)i * path_stride], 1e-8);
            double slope = -(evasion_sharpness / (v * v));
            if (pass == 0) {
                for (int k = i; k < horizon; ++k)
                    slope = std::min(slope, (upper[k] - gap - d) / ((double)std::max(1, std::abs(k - i)) * step));
            } else {
                for (int k = i; k >= 0; --k)
                    slope = std::min(slope, (upper[k] - gap - d) / ((double)std::max(1, std::abs(k - i)) * step));
                slope = std::min(slope, (proj_distance - d) / std::max(1.0, (double)i * step));
            }
            d += step * slope;
        }
    }
    for (int i = 0; i < n_path; ++i) out[i] = std::max(fwd[i], bwd[i]);
    return TPLHIP_OK;
}

// RefLine(RefLinePoint*, len, step_size, no_offset), utils.hpp:184-214: the bounding-box centre is subtracted
int tplhip_ref_line_center(const double* points, int n, int no_offset, double* out, double* x_offset, double* y_offset)
{
    if (!points || !out || !x_offset || !y_offset || n < 0) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    std::memcpy(out, points, sizeof(double) * 9 * (size_t)n);
    *x_offset = 0.0;
    *y_offset = 0.0;
    if (no_offset) return TPLHIP_OK;
    double x_min = INFINITY, x_max = -INFINITY, y_min = INFINITY, y_max = -INFINITY;
    for (int i = 0; i < n; ++i) {
        x_min = std::min(x_min, out[9 * i]);     x_max = std::max(x_max, out[9 * i]);
        y_min = std::min(y_min, out[9 * i + 1]); y_max = std::max(y_max, out[9 * i + 1]);
    }
    *x_offset = x_min + (x_max - x_min) / 2.0;
    *y_offset = y_min + (y_max - y_min) / 2.0;
    for (int i = 0; i < n; ++i) { out[9 * i] -= *x_offset; out[9 * i + 1] -= *y_offset; }
    return TPLHIP_OK;
}

// RefLine::lerp = interpRefLine, utils.hpp:148-173 (float weights promoted per product; the semantic channel is
// the nearer sample's)
int tplhip_ref_line_lerp(const double* points, int n, float step_size, float s, double* out)
{
    if (!points || !out || n < 1) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    float a = s / step_size;
    const int i0 = std::max(0, std::min(n - 1, (int)std::floor(a)));
    const int i1 = std::max(0, std::min(n - 1, (int)std::ceil(a)));
    a -= (float)i0;
    const float ia = (float)(1.0 - (double)a);
    const double* p0 = points + 9 * (size_t)i0;
    const double* p1 = points + 9 * (size_t)i1;
    for (int c = 0; c < 8; ++c) out[c] = p0[c] * (double)ia + p1[c] * (double)a;
    out[2] = p0[2] + tplhip::short_angle_dist(p0[2], p1[2]) * (double)a;
    out[8] = (ia < 0.5f) ? p0[8] : p1[8];
    return TPLHIP_OK;
}

// genPredictionGeometry, utils.cpp:576-693: the Frenet box of an object's hull on its map path, swept along the path
// for every prediction step, as triangle-soup vertices (x, y, t).  Two passes: out == nullptr (or cap too small)
// only counts; *n_out is always the number of vertices the call produces.
int tplhip_gen_prediction_geometry(const double* pred, int n_pred, const double* hull, int n_hull, const double* path,
                                   int n_path, const double* ts, int n_ts, double station_step_size,
                                   double expansion_rate, double sweep_length, double* out, int cap, int* n_out)
{
    if (!pred || !hull || !path || !ts || !n_out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (n_pred < 2) return tplhip_fail(TPLHIP_EINVAL, "a prediction needs at least two states (got %d)", n_pred);
    if (n_path < 2) return tplhip_fail(TPLHIP_EINVAL, "a line strip needs at least two points (got %d)", n_path);
    if (n_ts < 1) return tplhip_fail(TPLHIP_EINVAL, "ts is empty");
    const double pdx = path[2] - path[0], pdy = path[3] - path[1];
    const double path_step = std::sqrt(pdx * pdx + pdy * pdy);
    auto get_path = [&](double s, double* xy) {
        double a = s / path_step;
        const int i_prev = std::max(0, std::min(n_path - 2, (int)std::floor(a)));
        const int i_next = std::max(1, std::min(n_path - 1, (int)std::ceil(a)));
        a = std::max(0.0, std::min(1.0, a - i_prev));
        xy[0] = path[2 * i_prev] * (1.0 - a) + path[2 * i_next] * a;
        xy[1] = path[2 * i_prev + 1] * (1.0 - a) + path[2 * i_next + 1] * a;
    };
    // the hull as a box in Frenet coordinates of the path
    double s_min = INFINITY, s_max = -INFINITY, d_min = INFINITY, d_max = -INFINITY;
    for (int i = 0; i < n_hull; ++i) {
        TplProjection pr;
        project_one(path, 2, n_path, false, hull[2 * i], hull[2 * i + 1], &pr);
        if (pr.in_bounds) {
            s_min = std::min(pr.arc_len, s_min); s_max = std::max(pr.arc_len, s_max);
            d_min = std::min(pr.distance, d_min); d_max = std::max(pr.distance, d_max);
        }
    }
    double pp0[5];
    interp_prediction(pred, n_pred, ts[0], pp0);
    TplProjection proj0;
    project_one(path, 2, n_path, false, pp0[1], pp0[2], &proj0);
    double l = s_max - s_min;
    double s_mid = proj0.arc_len;
    long n = 0;
    double prev_min[3] = {0, 0, 0}, prev_max[3] = {0, 0, 0};       // the last two vertices pushed
    auto push = [&](double x, double y, double t) {
        if (out && n < cap) { out[3 * n] = x; out[3 * n + 1] = y; out[3 * n + 2] = t; }
        prev_min[0] = prev_max[0]; prev_min[1] = prev_max[1]; prev_min[2] = prev_max[2];
        prev_max[0] = x; prev_max[1] = y; prev_max[2] = t;
        ++n;
    };
    for (int t_idx = 0; t_idx < n_ts - 1; ++t_idx) {
        const double t = ts[t_idx], dt = ts[t_idx + 1] - t;
        double pp[5];
        interp_prediction(pred, n_pred, t, pp);
        const double sg = pp[4] < 0.0 ? -1.0 : 1.0;
        const double s_start = s_mid - sg * l * 0.5;
        const double s_stop = s_mid + std::max(sweep_length, dt) * pp[4] + sg * l * 0.5;
        // a hull with no vertex in bounds leaves the box at inf - inf: the reference's (int) of a non-finite double is
        // INT_MIN on x86-64 and nothing is drawn; made explicit here
        const double n_steps = std::abs(s_stop - s_start) / station_step_size + 1;
        const int steps = (std::isfinite(n_steps) && n_steps < 2147483648.0) ? (int)n_steps : 0;
        for (int i = 0; i < steps; ++i) {
            const double s = s_start + i * station_step_size * sg;
            const double ds = sg * std::min(std::abs(s_stop - s), station_step_size);
            if (std::abs(ds) < 1e-3) break;
            double p0[2], p1[2];
            get_path(s, p0);
            get_path(s + ds, p1);
            const double vx = (p1[0] - p0[0]) * sg, vy = (p1[1] - p0[1]) * sg;
            const double vl = std::sqrt(vx * vx + vy * vy);
            if (vl < 1e-3) break;
            const double ox = -vy / vl, oy = vx / vl;
            if (i == 0) {
                push(p0[0] + ox * d_min, p0[1] + oy * d_min, t);
                push(p1[0] + ox * d_min, p1[1] + oy * d_min, t);
                push(p0[0] + ox * d_max, p0[1] + oy * d_max, t);
                push(p0[0] + ox * d_max, p0[1] + oy * d_max, t);
                push(p1[0] + ox * d_min, p1[1] + oy * d_min, t);
                push(p1[0] + ox * d_max, p1[1] + oy * d_max, t);
            } else {
                // joined to the previous segment: its last two vertices (the t they carry included)
                const double a[3] = {prev_min[0], prev_min[1], prev_min[2]};
                const double b[3] = {prev_max[0], prev_max[1], prev_max[2]};
                const double cx = p1[0] + ox * d_min, cy = p1[1] + oy * d_min;
                push(a[0], a[1], a[2]);
                push(cx, cy, t);
                push(b[0], b[1], b[2]);
                push(b[0], b[1], b[2]);
                push(cx, cy, t);
                push(p1[0] + ox * d_max, p1[1] + oy * d_max, t);
            }
        }
        l *= 1.0 + expansion_rate;
        s_mid += dt * pp[4];
    }
    *n_out = (int)n;
    if (out && n > cap) return tplhip_fail(TPLHIP_EINVAL, "vertex buffer holds %d vertices, the geometry has %ld", cap, n);
    return TPLHIP_OK;
}

// tplcpp.lerp (binding utils.cpp:931-990; InterpVars(array, v, clipAlpha) utils.hpp:88-133): values ys[m][d] given at
// the sorted abscissae xs[m], evaluated at vals[n]; angle != 0 interpolates the short way round
int tplhip_lerp(const double* vals, int n, const double* xs, int m, const double* ys, int d, int angle, int clip_alpha,
                double* out)
{
    if (!vals || !xs || !ys || !out || d < 1) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (m < 2) return tplhip_fail(TPLHIP_EINVAL, "interpolation needs at least two samples (got %d)", m);
    for (int k = 0; k < n; ++k) {
        const double v = vals[k];
        long i_prev = 0, i_next = 0;
        if (v < xs[0]) { i_prev = 0; i_next = 1; }
        else if (v >= xs[m - 1]) { i_prev = m - 2; i_next = m - 1; }
        else if (m < 100) {
            for (long i = 1; i < m; ++i)
                if (xs[i - 1] <= v && v < xs[i]) { i_prev = i - 1; i_next = i; break; }
        } else {
            i_prev = 0; i_next = m - 1;
            while (i_next - i_prev >= 2) {
                const long mid = (i_next + i_prev) / 2;
                if (xs[mid] <= v) i_prev = mid;
                else if (xs[mid] > v) i_next = mid;
                else break;
            }
        }
        double a = (v - xs[i_prev]) / (xs[i_next] - xs[i_prev]);
        if (clip_alpha) a = std::max(0.0, std::min(1.0, a));
        for (int j = 0; j < d; ++j) {
            const double y0 = ys[i_prev * d + j], y1 = ys[i_next * d + j];
            out[(long)k * d + j] = angle ? y0 + a * tplhip::short_angle_dist(y0, y1) : (1.0 - a) * y0 + a * y1;
        }
    }
    return TPLHIP_OK;
}

// pointInPolygon, utils.cpp:25-65: the reference's edge-crossing count (its own handling of points on edges kept)
int tplhip_point_in_polygon(const double* points, int m, const double* polygon, int n, uint8_t* out)
{
    if (!points || !polygon || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (n < 1) return tplhip_fail(TPLHIP_EINVAL, "empty polygon");
    for (int k = 0; k < m; ++k) {
        const double x = points[2 * k], y = points[2 * k + 1];
        double p1x = polygon[0], p1y = polygon[1], xints = 0.0;
        bool inside = false;
        for (int i = 0; i < n + 1; ++i) {
            const double p2x = polygon[2 * (i % n)], p2y = polygon[2 * (i % n) + 1];
            if (std::min(p1y, p2y) <= y && y <= std::max(p1y, p2y) && x <= std::max(p1x, p2x)) {
                if (p1y != p2y) xints = (y - p1y) * (p2x - p1x) / (p2y - p1y) + p1x;
                if (p1x == p2x || x <= xints) inside = !inside;
            }
            p1x = p2x; p1y = p2y;
        }
        out[k] = inside ? 1 : 0;
    }
    return TPLHIP_OK;
}

// intersectPolygons, utils.cpp:87-158: separating-axis test over the edge normals of both (convex) polygons
int tplhip_intersect_polygons(const double* poly0, int l0, const double* poly1, int l1, int* out)
{
    if (!out || (!poly0 && l0 > 0) || (!poly1 && l1 > 0)) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    *out = 0;
    if (l0 <= 0 || l1 <= 0) return TPLHIP_OK;                 // an empty polygon intersects nothing
    for (int pass = 0; pass < 2; ++pass) {
        const double* p = pass == 0 ? poly0 : poly1;
        const int l = pass == 0 ? l0 : l1;
        for (int i = 0; i < l; ++i) {
            const double ax = p[2 * i], ay = p[2 * i + 1];
            double ex = p[2 * ((i + 1) % l)] - ax, ey = p[2 * ((i + 1) % l) + 1] - ay;
            const double en = std::sqrt(ex * ex + ey * ey);
            ex /= en; ey /= en;
            const double ox = -ey, oy = ex;
            double min0 = INFINITY, max0 = -INFINITY, min1 = INFINITY, max1 = -INFINITY;
            for (int j = 0; j < l0; ++j) {
                const double q = ox * (poly0[2 * j] - ax) + oy * (poly0[2 * j + 1] - ay);
                min0 = std::min(q, min0); max0 = std::max(q, max0);
            }
            for (int j = 0; j < l1; ++j) {
                const double q = ox * (poly1[2 * j] - ax) + oy * (poly1[2 * j + 1] - ay);
                min1 = std::min(q, min1); max1 = std::max(q, max1);
            }
            if ((max0 < min1 && max0 < max1) || (min0 > min1 && min0 > max1)) return TPLHIP_OK;   // separating axis
        }
    }
    *out = 1;
    return TPLHIP_OK;
}

// convexHull, utils.cpp:160-255: Graham scan from the lowest (then leftmost) point; out holds at most n points
int tplhip_convex_hull(const double* points, int n, double* out, int* n_out)
{
    if (!out || !n_out || (!points && n > 0)) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    struct P { double x, y; };
    std::vector<P> ps((size_t)std::max(n, 0));
    for (int i = 0; i < n; ++i) ps[(size_t)i] = {points[2 * i], points[2 * i + 1]};
    auto emit = [&](const std::vector<P>& v) {
        for (size_t i = 0; i < v.size(); ++i) { out[2 * i] = v[i].x; out[2 * i + 1] = v[i].y; }
        *n_out = (int)v.size();
        return TPLHIP_OK;
    };
    size_t len = ps.size();
    if (len < 4) return emit(ps);
    for (size_t i = 1; i < len; ++i) {
        if (ps[i].y > ps[0].y) continue;
        if (ps[i].y == ps[0].y && ps[i].x >= ps[0].x) continue;
        std::swap(ps[0], ps[i]);
    }
    auto orientation = [](const P& p, const P& q, const P& r) {
        const double res = (q.y - p.y) * (r.x - q.x) - (q.x - p.x) * (r.y - q.y);
        return res == 0 ? 0 : (res > 0 ? 1 : 2);              // colinear / clockwise / counter-clockwise
    };
    const P p0 = ps[0];
    std::sort(ps.begin() + 1, ps.end(), [&](const P& a, const P& b) {
        const int o = orientation(p0, a, b);
        if (o == 0) {
            const double d1 = (p0.x - a.x) * (p0.x - a.x) + (p0.y - a.y) * (p0.y - a.y);
            const double d2 = (p0.x - b.x) * (p0.x - b.x) + (p0.y - b.y) * (p0.y - b.y);
            return d1 < d2;
        }
        return o == 2;
    });
    size_t unique = 1;                                          // of colinear points keep the farthest
    for (size_t i = 1; i < len; i++) {
        while (i < len - 1 && orientation(p0, ps[i], ps[i + 1]) == 0) i++;
        ps[unique++] = ps[i];
    }
    ps.resize(unique);
    len = ps.size();
    if (len < 4) return emit(ps);
    std::vector<P> stack;
    stack.push_back(ps[0]);
    for (size_t i = 1; i < len; i++) {
        while (stack.size() > 1 && orientation(stack[stack.size() - 2], stack.back(), ps[i]) != 2) stack.pop_back();
        stack.push_back(ps[i]);
    }
    return emit(stack);
}

// smoothPath, utils.cpp:693-776: the path as the reference of an LQR over (x, x', x'', y, y', y'') with jerk inputs
// (lqrSmoother<6, 2>, utils.hpp:278-326, in plain loops); the end states are pinned with weight 1e6
int tplhip_smooth_path(const double* path, int n, double ds, double w_v, double w_a, double w_j, int closed, double* out)
{
    if (!path || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (n < 2) return tplhip_fail(TPLHIP_EINVAL, "a line strip needs at least two points (got %d)", n);
    constexpr int X = 6, U = 2;
    const size_t H = (size_t)n;
    std::vector<double> xref(H * X, 0.0);
    for (size_t i = 0; i < H; ++i) { xref[i * X + 0] = path[2 * i]; xref[i * X + 3] = path[2 * i + 1]; }
    {
        double dx = path[2] - path[0], dy = path[3] - path[1];
        const double dn = std::sqrt(dx * dx + dy * dy);
        dx = dx / dn * ds; dy = dy / dn * ds;
        xref[1] = dx; xref[4] = dy;
    }
    if (closed) for (int c = 0; c < X; ++c) xref[(H - 1) * X + c] = xref[c];
    double A[X][X] = {}, B[X][U] = {};
    for (int i = 0; i < X; ++i) A[i][i] = 1.0;
    A[0][1] = ds; A[1][2] = ds; A[3][4] = ds; A[4][5] = ds;
    B[2][0] = ds; B[5][1] = ds;
    const double qd[X] = {1.0, w_v, w_a, 1.0, w_v, w_a};
    auto q_at = [&](size_t i, int c) { return (i == 0 || i == H - 1) ? 1.0e6 : qd[c]; };     // Qs are diagonal
    std::vector<double> kf(H * U, 0.0), Kf(H * U * X, 0.0);
    double Vxx[X][X] = {}, Vx[X];
    for (int c = 0; c < X; ++c) { Vxx[c][c] = q_at(H - 1, c); Vx[c] = -q_at(H - 1, c) * xref[(H - 1) * X + c]; }
    for (long i = (long)H - 2; i >= 0; --i) {
        double lx[X], Qx[X], Qu[U], Qxx[X][X], Quu[U][U], Qux[U][X], VA[X][X], VB[X][U];
        for (int c = 0; c < X; ++c) lx[c] = -q_at((size_t)i, c) * xref[(size_t)i * X + c];
        for (int r = 0; r < X; ++r) {                          // Vxx * fx, Vxx * fu
            for (int c = 0; c < X; ++c) { double a = 0; for (int k = 0; k < X; ++k) a += Vxx[r][k] * A[k][c]; VA[r][c] = a; }
            for (int c = 0; c < U; ++c) { double a = 0; for (int k = 0; k < X; ++k) a += Vxx[r][k] * B[k][c]; VB[r][c] = a; }
        }
        for (int r = 0; r < X; ++r) { double a = 0; for (int k = 0; k < X; ++k) a += A[k][r] * Vx[k]; Qx[r] = lx[r] + a; }
        for (int r = 0; r < U; ++r) { double a = 0; for (int k = 0; k < X; ++k) a += B[k][r] * Vx[k]; Qu[r] = a; }
        for (int r = 0; r < X; ++r)
            for (int c = 0; c < X; ++c) {
                double a = 0; for (int k = 0; k < X; ++k) a += A[k][r] * VA[k][c];
                Qxx[r][c] = (r == c ? q_at((size_t)i, r) : 0.0) + a;
            }
        for (int r = 0; r < U; ++r) {
            for (int c = 0; c < U; ++c) { double a = 0; for (int k = 0; k < X; ++k) a += B[k][r] * VB[k][c]; Quu[r][c] = (r == c ? w_j : 0.0) + a; }
            for (int c = 0; c < X; ++c) { double a = 0; for (int k = 0; k < X; ++k) a += B[k][r] * VA[k][c]; Qux[r][c] = a; }
        }
        const double det = Quu[0][0] * Quu[1][1] - Quu[0][1] * Quu[1][0];
        const double W[U][U] = {{-Quu[1][1] / det, Quu[0][1] / det}, {Quu[1][0] / det, -Quu[0][0] / det}};   // -Quu^-1
        double* k = &kf[(size_t)i * U];
        double* K = &Kf[(size_t)i * U * X];
        for (int r = 0; r < U; ++r) {
            k[r] = W[r][0] * Qu[0] + W[r][1] * Qu[1];
            for (int c = 0; c < X; ++c) K[r * X + c] = W[r][0] * Qux[0][c] + W[r][1] * Qux[1][c];
        }
        double Quuk[U], QuuK[U][X];
        for (int r = 0; r < U; ++r) {
            Quuk[r] = Quu[r][0] * k[0] + Quu[r][1] * k[1];
            for (int c = 0; c < X; ++c) QuuK[r][c] = Quu[r][0] * K[c] + Quu[r][1] * K[X + c];
        }
        for (int r = 0; r < X; ++r)
            Vx[r] = Qx[r] + (K[r] * Quuk[0] + K[X + r] * Quuk[1]) + (K[r] * Qu[0] + K[X + r] * Qu[1])
                    + (Qux[0][r] * k[0] + Qux[1][r] * k[1]);
        for (int r = 0; r < X; ++r)
            for (int c = 0; c < X; ++c) {
                const double t1 = K[r] * Qux[0][c] + K[X + r] * Qux[1][c];
                const double t2 = K[c] * Qux[0][r] + K[X + c] * Qux[1][r];
                Vxx[r][c] = Qxx[r][c] + (K[r] * QuuK[0][c] + K[X + r] * QuuK[1][c]) + t1 + t2;
            }
    }
    double x[X];
    for (int c = 0; c < X; ++c) x[c] = xref[c];
    out[0] = x[0]; out[1] = x[3];
    for (size_t i = 0; i + 1 < H; ++i) {
        const double* k = &kf[i * U];
        const double* K = &Kf[i * U * X];
        double u[U], xn[X];
        for (int r = 0; r < U; ++r) { double a = 0; for (int c = 0; c < X; ++c) a += K[r * X + c] * x[c]; u[r] = a + k[r]; }
        for (int r = 0; r < X; ++r) { double a = 0; for (int c = 0; c < X; ++c) a += A[r][c] * x[c]; xn[r] = a + B[r][0] * u[0] + B[r][1] * u[1]; }
        for (int c = 0; c < X; ++c) x[c] = xn[c];
        out[2 * (i + 1)] = x[0]; out[2 * (i + 1) + 1] = x[3];
    }
    return TPLHIP_OK;
}

} // extern "C"
