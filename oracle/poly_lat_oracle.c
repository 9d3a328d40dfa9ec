/*
 * poly_lat_oracle.c -- CPU restatement of tpl's lateral quintic-lattice planner.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tpl_amd/ may include, link or call this file; it is
 * the checker for the HIP path (tests/), never the thing shipped or measured.
 *
 * PARITY: the quintic (PolyQuintic) is PINNED against the reference's own source --
 * library/src/poly_interp.cu compiles as host C++ and is built into oracle/_ref/ by
 * oracle/Makefile (tests/test_poly_lat_oracle.py compares the two).  The planner kernels are
 * CUDA with texture/surface objects and cannot be built here: PARITY UNPINNED for them; this
 * file restates their source semantics kernel by kernel:
 *
 *   PolyQuintic                 library/src/poly_interp.cu:96-168, 384-416
 *   evalLocalPathCosts          library/src/dyn_prog/poly_lat_planner.cu:10-103
 *   calcPathTimes               poly_lat_planner.cu:105-147
 *   checkCollisions             poly_lat_planner.cu:149-185
 *   evalPathCosts               poly_lat_planner.cu:187-222
 *   selectPath                  poly_lat_planner.cu:224-266
 *   PolyLatTraj::lerp / updateTimeDistCurv      poly_lat_planner.cu:270-320
 *   PolyLatPlanner::update (driver, final path) poly_lat_planner.cu:354-486
 *   interpRefLine, shortAngleDist, InterpVars   library/include/tplcpp/utils.hpp:21-173
 *   interpDistField             library/src/dyn_prog/env.cu:225-234
 *
 * Rules: R1-R3 of dp_lon_oracle.c (C++ promotions of the source, no FMA contraction, texture
 * filter with 8-bit weights, blend order z, y, x).  float overloads of the device code
 * (atan / cos / sqrt of a float argument) are atanf / cosf / sqrtf.  The reference reads one
 * element past the end for the last point's curvature in updateTimeDistCurv (undefined); it is
 * 0 here.
 */
#define _GNU_SOURCE
#include "../include/tplhip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PLO_API __attribute__((visibility("default")))

typedef struct { double x, y, heading, s, k, v_max, d_left, d_right, semantic; } RefPt;

static int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static double sqd(double x) { return x * x; }
static float sqf(float x) { return x * x; }

/* ---- utils.hpp:21-53, double and float instantiations --------------------------------- */

static double normalize_angle(double x)
{
    x = fmod(x, 2 * M_PI);
    x = fmod(x + 2 * M_PI, 2 * M_PI);
    if (x > M_PI) x -= 2 * M_PI;
    return x;
}

static double short_angle_dist(double x, double y)
{
    x = normalize_angle(x);
    y = normalize_angle(y);
    double a0 = y - x, a1 = y - x + 2 * M_PI, a2 = y - x - 2 * M_PI;
    if (fabs(a1) < fabs(a0)) a0 = a1;
    if (fabs(a2) < fabs(a0)) a0 = a2;
    return a0;
}

static float normalize_angle_f(float x)
{
    x = fmod(x, 2 * M_PI);
    x = fmod(x + 2 * M_PI, 2 * M_PI);
    if (x > M_PI) x -= 2 * M_PI;
    return x;
}

static float short_angle_dist_f(float x, float y)
{
    x = normalize_angle_f(x);
    y = normalize_angle_f(y);
    float a0 = y - x;
    float a1 = y - x + 2 * M_PI;
    float a2 = y - x - 2 * M_PI;
    if (fabsf(a1) < fabsf(a0)) a0 = a1;
    if (fabsf(a2) < fabsf(a0)) a0 = a2;
    return a0;
}

/* utils.hpp:148-173 */
static RefPt interp_ref_line(const RefPt* rl, int len, float step_size, float s)
{
    float a = s / step_size;
    int i0 = clampi((int)floorf(a), 0, len - 1);
    int i1 = clampi((int)ceilf(a), 0, len - 1);
    a -= i0;
    float ia = 1.0 - a;
    RefPt p;
    memset(&p, 0, sizeof p);
    p.x = rl[i0].x * ia + rl[i1].x * a;
    p.y = rl[i0].y * ia + rl[i1].y * a;
    p.heading = rl[i0].heading + short_angle_dist(rl[i0].heading, rl[i1].heading) * a;
    p.s = rl[i0].s * ia + rl[i1].s * a;
    p.k = rl[i0].k * ia + rl[i1].k * a;
    p.v_max = rl[i0].v_max * ia + rl[i1].v_max * a;
    p.d_left = rl[i0].d_left * ia + rl[i1].d_left * a;
    p.d_right = rl[i0].d_right * ia + rl[i1].d_right * a;
    p.semantic = (ia < 0.5) ? rl[i0].semantic : rl[i1].semantic;
    return p;
}

/* ---- PolyQuintic, poly_interp.cu ------------------------------------------------------- */

PLO_API void plo_quintic_init(TplPolyQuintic* q, double x0, double y0, double dy0, double ddy0,
                              double x1, double y1, double dy1, double ddy1)
{
    q->x0 = x0; q->x1 = x1; q->d = x1 - x0;
    /* calc_poly5_coeffs(y0 / d, dy0, ddy0 * d, y1 / d, dy1, ddy1 * d, c), :96-108 */
    const double y0s = y0 / q->d, y0d = dy0, y0dd = ddy0 * q->d, y1s = y1 / q->d, y1d = dy1, y1dd = ddy1 * q->d;
    const double x0_ = (1.0 / 2.0) * y0dd;
    const double x1_ = (3.0 / 2.0) * y0dd;
    const double x2_ = (1.0 / 2.0) * y1dd;
    q->c[0] = y0s;
    q->c[1] = y0d;
    q->c[2] = x0_;
    q->c[3] = -x1_ + x2_ - 10 * y0s - 6 * y0d + 10 * y1s - 4 * y1d;
    q->c[4] = x1_ + 15 * y0s + 8 * y0d - 15 * y1s + 7 * y1d - y1dd;
    q->c[5] = -x0_ + x2_ - 6 * y0s - 3 * y0d + 6 * y1s - 3 * y1d;
}

PLO_API double plo_quintic_eval(const TplPolyQuintic* q, int order, double x)
{
    const double* c = q->c;
    double t = (x - q->x0) / q->d;
    if (order == 0) return (t * (t * (t * (t * (t * c[5] + c[4]) + c[3]) + c[2]) + c[1]) + c[0]) * q->d;   /* :120 */
    if (order == 1) return t * (t * (t * (5 * t * c[5] + 4 * c[4]) + 3 * c[3]) + 2 * c[2]) + c[1];          /* :135 */
    if (order == 2) return (t * (t * (20 * t * c[5] + 12 * c[4]) + 6 * c[3]) + 2 * c[2]) / q->d;            /* :150 */
    return (t * (60 * t * c[5] + 24 * c[4]) + 6 * c[3]) / (q->d * q->d);                                   /* :165 */
}

/* ---- environment look-ups ------------------------------------------------------------ */

typedef struct { int i0, i1; float w; } Tap;

static Tap make_tap(float x_plus_half, int n)
{
    Tap t;
    float xb = x_plus_half - 0.5f;
    float fl = floorf(xb);
    float alpha = xb - fl;
    t.w = rintf(alpha * 256.0f) / 256.0f;
    float lo = fminf(fmaxf(fl, -1.0f), (float)n);
    int i = (int)lo;
    t.i0 = clampi(i, 0, n - 1);
    t.i1 = clampi(i + 1, 0, n - 1);
    return t;
}

static float lerp1(float p, float q, float w) { return fmaf(q, w, p * (1.0f - w)); }

/* interpDistField, env.cu:225-234; occ is float[l][s][t] */
static float dist_field(const TplDpEnvParams* ep, const float* occ, float t, float s, float l)
{
    float t_idx = t < ep->dt_start ? 0 : (roundf((t - ep->dt_start) / ep->dt) + 1.0);
    float sc = (s - ep->s_min) / (ep->s_max - ep->s_min) * (ep->s_steps - 1);
    float lc = (l - ep->l_min) / (ep->l_max - ep->l_min) * (ep->l_steps - 1);
    Tap tt = make_tap(t_idx + 0.5f, ep->t_steps);
    Tap ts = make_tap(sc + 0.5f, ep->s_steps);
    Tap tl = make_tap(lc + 0.5f, ep->l_steps);
#define AT(il, is, it) occ[((size_t)(il) * ep->s_steps + (is)) * ep->t_steps + (it)]
    float b00 = lerp1(AT(tl.i0, ts.i0, tt.i0), AT(tl.i1, ts.i0, tt.i0), tl.w);
    float b10 = lerp1(AT(tl.i0, ts.i1, tt.i0), AT(tl.i1, ts.i1, tt.i0), tl.w);
    float b01 = lerp1(AT(tl.i0, ts.i0, tt.i1), AT(tl.i1, ts.i0, tt.i1), tl.w);
    float b11 = lerp1(AT(tl.i0, ts.i1, tt.i1), AT(tl.i1, ts.i1, tt.i1), tl.w);
#undef AT
    return lerp1(lerp1(b00, b10, ts.w), lerp1(b01, b11, ts.w), tt.w);
}

/* ---- the lattice ------------------------------------------------------------------------ */

#define NODE(il, is, s) (nodes + 4 * (((size_t)(il) * p->s_dst_steps + (is)) * p->s_steps + (s)))

/* Runs the reference's five kernels in order.  nodes: float[l_dst][s_dst][s][4]; best: [2]. */
PLO_API void plo_lattice(const TplPolyLatParams* p, const TplDpEnvParams* ep, const TplPolyLatTrajPoint* start,
                         const double* ref_centered, int n_ref, float ref_step, const float* occ,
                         float* nodes, int* best)
{
    const RefPt* rl = (const RefPt*)ref_centered;

    /* evalLocalPathCosts */
    for (int idx_l_dst = 0; idx_l_dst < p->l_dst_steps; ++idx_l_dst)
        for (int idx_s_dst = 0; idx_s_dst < p->s_dst_steps; ++idx_s_dst)
            for (int idx_s = 0; idx_s < p->s_steps; ++idx_s) {
                const float l_dst = p->l_dst_min + p->_l_dst_step_size * idx_l_dst;
                const float s_dst = p->s_dst_min + p->_s_dst_step_size * idx_s_dst;
                const float s = start->s + p->s_min + p->_s_step_size * idx_s;
                TplPolyQuintic poly;
                plo_quintic_init(&poly, start->s, start->l, start->dl, start->ddl, start->s + s_dst, l_dst, 0.0, 0.0);
                float pl, pdl, pddl, pdddl;
                if (s >= start->s + s_dst) {
                    pl = l_dst; pdl = 0.0; pddl = 0.0; pdddl = 0.0;
                } else {
                    pl = plo_quintic_eval(&poly, 0, s);
                    pdl = plo_quintic_eval(&poly, 1, s);
                    pddl = plo_quintic_eval(&poly, 2, s);
                    pdddl = plo_quintic_eval(&poly, 3, s);
                }
                const RefPt rp = interp_ref_line(rl, n_ref, ref_step, s);
                const float heading_frenet = atanf(pdl);
                float pk = (pddl / (pdl * pdl + 1.0) + rp.k) * cosf(heading_frenet) / (1.0 - pl * rp.k);
                float k_abs_path = fmax(fabsf(pk), fabs(rp.k));
                float pv = rp.v_max;
                if (k_abs_path > 1e-6) pv = fminf(pv, sqrtf(p->a_lat_abs_max / k_abs_path));

                float constr = 0.0;
                constr += fmax(0.0, sqd(fmin(rp.v_max, start->v)) * fabsf(pk) - p->a_lat_abs_max);
                if (s <= start->s + s_dst) {
                    constr += fmax(0.0, fabsf(pk) - p->k_abs_max);
                    constr += fmax(0.0, pl - (rp.d_left - p->width_veh * 0.5 * sqrt(2.0)));
                    constr += fmax(0.0, (-rp.d_right + p->width_veh * 0.5 * sqrt(2.0)) - pl);
                }
                float cost = 0.0;
                cost += p->w_dl * sqf(pdl);
                cost += p->w_ddl * sqf(pddl);
                cost += p->w_dddl * sqf(pdddl);
                if (fabsf(pk) > fabsf((float)rp.k)) cost += p->w_k * sqf(pk);      /* fabsf(double) converts first */
                cost += 10e6 * constr;
                float* n = NODE(idx_l_dst, idx_s_dst, idx_s);
                n[0] = pl; n[1] = pv; n[2] = 0.0; n[3] = cost;
            }

    /* calcPathTimes */
    for (int idx_l_dst = 0; idx_l_dst < p->l_dst_steps; ++idx_l_dst)
        for (int idx_s_dst = 0; idx_s_dst < p->s_dst_steps; ++idx_s_dst) {
            float dist = 0.0, t = 0.0, x_prev = 0.0, y_prev = 0.0;
            for (int idx_s = 0; idx_s < p->s_steps; ++idx_s) {
                const float s = start->s + p->s_min + p->_s_step_size * idx_s;
                const RefPt rp = interp_ref_line(rl, n_ref, ref_step, s);
                float* n = NODE(idx_l_dst, idx_s_dst, idx_s);
                const float l = n[0];
                const float x = rp.x - sin(rp.heading) * l;
                const float y = rp.y + cos(rp.heading) * l;
                const float diff_x = x - x_prev, diff_y = y - y_prev;
                if (idx_s > 0) {
                    const float d = sqrtf(diff_x * diff_x + diff_y * diff_y);
                    dist += d;
                    t += d / fmax(1.0, n[1]);
                }
                x_prev = x; y_prev = y;
                n[2] = t;
            }
            (void)dist;
        }

    /* checkCollisions */
    for (int idx_l_dst = 0; idx_l_dst < p->l_dst_steps; ++idx_l_dst)
        for (int idx_s_dst = 0; idx_s_dst < p->s_dst_steps; ++idx_s_dst)
            for (int idx_s = 0; idx_s < p->s_steps; ++idx_s) {
                const float s = start->s + p->s_min + p->_s_step_size * idx_s;
                float* n = NODE(idx_l_dst, idx_s_dst, idx_s);
                const float l = n[0], t = n[2];
                float dist_sem = 0.0;
                for (float t_sweep = -1.0; t_sweep < 1.1; t_sweep += 1.0) {
                    dist_sem = fmaxf(dist_sem, dist_field(ep, occ, t + t_sweep, s, l));
                    dist_sem = fmaxf(dist_sem, dist_field(ep, occ, t + t_sweep, s, l + 0.25));
                    dist_sem = fmaxf(dist_sem, dist_field(ep, occ, t + t_sweep, s, l - 0.25));
                }
                int collision = dist_sem > 0.0;
                n[2] = (collision && t < 8.0 && s > p->length_veh) ? s : 10000.0;
            }

    /* evalPathCosts */
    for (int idx_l_dst = 0; idx_l_dst < p->l_dst_steps; ++idx_l_dst)
        for (int idx_s_dst = 0; idx_s_dst < p->s_dst_steps; ++idx_s_dst) {
            const float l_dst = p->l_dst_min + p->_l_dst_step_size * idx_l_dst;
            const float s_dst = p->s_dst_min + p->_s_dst_step_size * idx_s_dst;
            float traj_cost = 0.0, collision_dist = 1000.0;
            for (int idx_s = 0; idx_s < p->s_steps; ++idx_s) {
                const float* n = NODE(idx_l_dst, idx_s_dst, idx_s);
                traj_cost += n[3];
                collision_dist = fminf(collision_dist, n[2]);
            }
            if (l_dst < -0.1) traj_cost += p->w_right;
            float diff_l = l_dst - p->l_trg;
            traj_cost += p->w_l * diff_l * diff_l;
            traj_cost += p->w_len * fabsf(s_dst);
            float* n0 = NODE(idx_l_dst, idx_s_dst, 0);
            n0[2] = collision_dist;
            n0[3] = traj_cost;
        }

    /* selectPath */
    {
        int min_idx_l_dst = p->l_dst_steps / 2 + 1;
        int min_idx_s_dst = p->s_dst_steps - 1;
        const float* center = NODE(clampi(min_idx_l_dst, 0, p->l_dst_steps - 1), min_idx_s_dst, 0);
        float max_collision_dist = center[2];
        float min_cost = INFINITY;
        for (int il = 0; il < p->l_dst_steps; ++il)
            for (int is = 0; is < p->s_dst_steps; ++is) {
                const float* n = NODE(il, is, 0);
                if (n[3] >= 1e6) continue;
                if (n[2] > max_collision_dist + p->length_veh) max_collision_dist = n[2];
            }
        for (int il = 0; il < p->l_dst_steps; ++il)
            for (int is = 0; is < p->s_dst_steps; ++is) {
                const float* n = NODE(il, is, 0);
                if (fabs(n[2] - max_collision_dist) > 1.0) continue;
                if (n[3] < min_cost) {
                    min_cost = n[3];
                    max_collision_dist = n[2];
                    min_idx_l_dst = il;
                    min_idx_s_dst = is;
                }
            }
        best[0] = min_idx_l_dst;
        best[1] = min_idx_s_dst;
    }
}

/* PolyLatTraj::updateTimeDistCurv, :300-320 */
PLO_API void plo_traj_update(TplPolyLatTrajPoint* points, int n)
{
    double dist = 0.0, t = 0.0;
    for (int i = 1; i < n + 1; ++i) {
        TplPolyLatTrajPoint* p = &points[i - 1];
        p->distance = dist;
        p->t = t;
        if (i == n) { p->k = 0.0f; break; }            /* undefined in the reference */
        const TplPolyLatTrajPoint* pn = &points[i];
        double dx = pn->x - p->x, dy = pn->y - p->y;
        double d = sqrt(dx * dx + dy * dy);
        p->k = short_angle_dist_f(p->heading, pn->heading) / d;
        dist += d;
        t += d / p->v;
    }
}

/* PolyLatTraj::lerp, :270-295 */
PLO_API void plo_traj_lerp(const TplPolyLatTrajPoint* pts, int n, double distance, TplPolyLatTrajPoint* res)
{
    int i_prev = 0, i_next = 0;
    if (distance < pts[0].distance) { i_prev = 0; i_next = 1; }
    else if (distance >= pts[n - 1].distance) { i_prev = n - 2; i_next = n - 1; }
    else {
        for (int i = 1; i < n; ++i)
            if (pts[i - 1].distance <= distance && distance < pts[i].distance) { i_prev = i - 1; i_next = i; break; }
    }
    double a = (distance - pts[i_prev].distance) / (pts[i_next].distance - pts[i_prev].distance);
    a = fmax(0.0, fmin(1.0, a));
    double ai = 1.0f - a;
    memset(res, 0, sizeof *res);
    res->t = pts[i_prev].t * ai + pts[i_next].t * a;
    res->l = pts[i_prev].l * ai + pts[i_next].l * a;
    res->dl = pts[i_prev].dl * ai + pts[i_next].dl * a;
    res->ddl = pts[i_prev].ddl * ai + pts[i_next].ddl * a;
    res->dddl = pts[i_prev].dddl * ai + pts[i_next].dddl * a;
    res->s = pts[i_prev].s * ai + pts[i_next].s * a;
    res->v = pts[i_prev].v * ai + pts[i_next].v * a;
    res->x = pts[i_prev].x * (double)ai + pts[i_next].x * (double)a;
    res->y = pts[i_prev].y * (double)ai + pts[i_next].y * (double)a;
    res->heading = pts[i_prev].heading + short_angle_dist_f(pts[i_prev].heading, pts[i_next].heading) * a;
    res->distance = pts[i_prev].distance * ai + pts[i_next].distance * a;
    res->k = pts[i_prev].k * ai + pts[i_next].k * a;
}

/* final path of PolyLatPlanner::update, :441-483 */
PLO_API void plo_final_path(const TplPolyLatParams* p, const TplPolyLatTrajPoint* start, const double* ref_centered,
                            int n_ref, float ref_step, const double* offsets, const int* best,
                            TplPolyLatTrajPoint* points, TplPolyQuintic* poly)
{
    const RefPt* rl = (const RefPt*)ref_centered;
    const float l_dst = p->l_dst_min + p->_l_dst_step_size * best[0];
    const float s_dst = p->s_dst_min + p->_s_dst_step_size * best[1];
    plo_quintic_init(poly, start->s, start->l, start->dl, start->ddl, start->s + s_dst, l_dst, 0.0, 0.0);
    for (int idx_s = 0; idx_s < p->s_steps; ++idx_s) {
        TplPolyLatTrajPoint* pt = &points[idx_s];
        memset(pt, 0, sizeof *pt);
        pt->s = start->s + p->s_min + p->_s_step_size * idx_s;
        if (pt->s >= start->s + s_dst) {
            pt->l = l_dst; pt->dl = 0.0; pt->ddl = 0.0; pt->dddl = 0.0;
        } else {
            pt->l = plo_quintic_eval(poly, 0, pt->s);
            pt->dl = plo_quintic_eval(poly, 1, pt->s);
            pt->ddl = plo_quintic_eval(poly, 2, pt->s);
            pt->dddl = plo_quintic_eval(poly, 3, pt->s);
        }
        const RefPt rp = interp_ref_line(rl, n_ref, ref_step, pt->s);
        const float heading_frenet = atanf(pt->dl);
        pt->x = offsets[0] + rp.x - sin(rp.heading) * pt->l;
        pt->y = offsets[1] + rp.y + cos(rp.heading) * pt->l;
        pt->heading = heading_frenet + rp.heading;
        pt->v = rp.v_max;
    }
    plo_traj_update(points, p->s_steps);
}
