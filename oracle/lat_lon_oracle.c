/*
 * lat_lon_oracle.c -- CPU restatement of tpl's DP (s, ds, l) lateral/longitudinal planner
 * (SURVEY 8 f-3).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tpl_amd/ may include, link or call this file; it is
 * the checker for the HIP path (tests/), never the thing shipped or measured.
 *
 * PARITY UNPINNED: the reference is CUDA (3-D textures/surfaces) plus Eigen for the smoother;
 * neither can be built here and the reference's only test of this planner (library/test/
 * test_sim.py) is a closed-loop rule check that needs the whole simulator.  This file restates
 * the source semantics:
 *
 *   dynamics, EvalResult, Evaluator (finalState, getMid, evalState, evalAction,
 *   findBestAction), backwardStep, forwardStep, reevalTraj
 *                               library/src/dyn_prog/lat_lon_planner.cu:10-404
 *   LatLonTraj::state / lerp, LatLonCartTraj::lerp      lat_lon_planner.cu:432-486
 *   updateTrajDp (driver order)                         lat_lon_planner.cu:579-643
 *   updateTrajSmooth + lqrSmoother<4,1>                 lat_lon_planner.cu:645-760, utils.hpp:278-348
 *   frenetToCartesian                                   lat_lon_planner.cu:768-817
 *   interpDistMapLon, texLerp                           env.cu:236-252, common.cuh:93-95
 *   get_cached_scaled / lerp_cached_scaled              arr_tex_surf.cuh:151-167
 *
 * Rules: R1-R3 of dp_lon_oracle.c; a point-filtered texture fetch at unnormalised coordinate x
 * reads texel floor(x) (so get_cached_scaled reads floor((v - min)/step + 0.5)), clamped.
 * The products of the LQR recursion associate left to right as written in the source, each
 * small matrix product summing its inner index in ascending order (Eigen's order for these
 * sizes is not specified: the smoother is compared at 1e-9, not bit for bit).
 */
#define _GNU_SOURCE
#include "../include/tplhip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LLO_API __attribute__((visibility("default")))

typedef struct { double x, y, heading, s, k, v_max, d_left, d_right, semantic; } RefPt;
typedef struct { float cost, constr; uint8_t flags; } EvalResult;
typedef struct { int i0, i1; float w; } Tap;
typedef struct { float x, y, z, w; } f4;

static int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static float sq(float x) { return x * x; }
/* The texture filter's frozen choices as RUN-TIME state (round 6; build switches until round 5), exactly those of
 * oracle/dp_lon_oracle.c (rules R2, R3 and R3's order):
 *   quant  1 rint (default) | 2 floor | 0 none            lerp  0 fma(q, w, p(1-w)) (default) | 1 | 2
 *   order  0 z -> y -> x (default: the slowest axis first) | 1 x -> y -> z | 2 eight-corner sum, fp32 | 3 the sum, wide
 * They act on every filtered fetch of this planner: the value tables (x = s, y = ds, z = l), the longitudinal distance map
 * (x = t, y = s, z = l) and the 1-D reference-line texture.  llo_set_rule selects the rule for all later calls of this library
 * (process-wide); tplhip_lat_lon_set_filter_rule is the same choice on the HIP side.  The file is built -ffp-contract=off:
 * every operation below keeps its own rounding. */
typedef struct { int quant, lerp, order; } LloRule;
static LloRule g_rule = { 1, 0, 0 };

LLO_API int llo_set_rule(int quant, int lerp, int order)
{
    if (quant < 0 || quant > 2 || lerp < 0 || lerp > 2 || order < 0 || order > 3) return -1;
    g_rule.quant = quant; g_rule.lerp = lerp; g_rule.order = order;
    return 0;
}

LLO_API void llo_get_rule(int out[3]) { out[0] = g_rule.quant; out[1] = g_rule.lerp; out[2] = g_rule.order; }

static float lerp1(float p, float q, float w)
{
    if (g_rule.lerp == 1) { const float a = p * (1.0f - w), b = q * w; return a + b; }
    if (g_rule.lerp == 2) { const float d = q - p; const float e = w * d; return p + e; }
    return fmaf(q, w, p * (1.0f - w));
}

/* sum of w[c] t[c] in index order: fp32 accumulator (every product and sum rounded) or exact products in a double accumulator */
static float wsum(const float* w, const float* t, int n)
{
    if (g_rule.order == 3) {
        double acc = (double)w[0] * (double)t[0];
        for (int c = 1; c < n; ++c) acc += (double)w[c] * (double)t[c];
        return (float)acc;
    }
    float acc = w[0] * t[0];
    for (int c = 1; c < n; ++c) { const float pr = w[c] * t[c]; acc = acc + pr; }
    return acc;
}

/* rule R3 over the eight corners t[x + 2 y + 4 z] with the weights of the three axes */
static float blend8(const float* t, float wx, float wy, float wz)
{
    if (g_rule.order == 0) {
        float lo = lerp1(lerp1(t[0], t[4], wz), lerp1(t[2], t[6], wz), wy);          /* z, then y, then x */
        float hi = lerp1(lerp1(t[1], t[5], wz), lerp1(t[3], t[7], wz), wy);
        return lerp1(lo, hi, wx);
    }
    if (g_rule.order == 1) {
        float y0 = lerp1(lerp1(t[0], t[1], wx), lerp1(t[2], t[3], wx), wy);          /* x, then y, then z */
        float y1 = lerp1(lerp1(t[4], t[5], wx), lerp1(t[6], t[7], wx), wy);
        return lerp1(y0, y1, wz);
    }
    const float x2[2] = {1.0f - wx, wx}, y2[2] = {1.0f - wy, wy}, z2[2] = {1.0f - wz, wz};
    float w[8];
    for (int c = 0; c < 8; ++c) { const float xy = x2[c & 1] * y2[(c >> 1) & 1]; w[c] = xy * z2[c >> 2]; }
    return wsum(w, t, 8);
}

static Tap make_tap(float x_plus_half, int n)
{
    Tap t;
    float xb = x_plus_half - 0.5f;
    float fl = floorf(xb);
    float alpha = xb - fl;
    if (g_rule.quant == 1)      t.w = rintf(alpha * 256.0f) / 256.0f;
    else if (g_rule.quant == 2) t.w = floorf(alpha * 256.0f) / 256.0f;
    else                        t.w = alpha;
    int i = (int)fminf(fmaxf(fl, -1.0f), (float)n);
    t.i0 = clampi(i, 0, n - 1);
    t.i1 = clampi(i + 1, 0, n - 1);
    return t;
}

/* everything a kernel sees */
typedef struct {
    const TplLatLonParams* p;
    const TplDpEnvParams* ep;
    const float* dist_lon;        /* [l][s][t][2] */
    const f4* ref_tex;            /* (v_max, d_left, d_right, semantic) as float, common.cuh:59-65 */
    int n_ref;
    float ref_step;
    float step_s, step_ds, step_l;   /* ArrTexSurf::rescale */
} Ctx;

/* interpDistMapLon(t, s, l).x */
static float dist_lon_x(const Ctx* c, float t, float s, float l)
{
    const TplDpEnvParams* ep = c->ep;
    float t_idx = t < ep->dt_start ? 0 : (roundf((t - ep->dt_start) / ep->dt) + 1.0);
    float sc = (s - ep->s_min) / (ep->s_max - ep->s_min) * (ep->s_steps - 1);
    float lc = roundf((l - ep->l_min) / (ep->l_max - ep->l_min) * (ep->l_steps - 1));
    Tap tt = make_tap(t_idx + 0.5f, ep->t_steps), ts = make_tap(sc + 0.5f, ep->s_steps), tl = make_tap(lc + 0.5f, ep->l_steps);
#define AT(il, is, it) c->dist_lon[((((size_t)(il)) * ep->s_steps + (is)) * ep->t_steps + (it)) * 2]
    /* corners x + 2 y + 4 z with x = t, y = s, z = l (default order: l, then s, then t) */
    const float t8[8] = { AT(tl.i0, ts.i0, tt.i0), AT(tl.i0, ts.i0, tt.i1), AT(tl.i0, ts.i1, tt.i0), AT(tl.i0, ts.i1, tt.i1),
                          AT(tl.i1, ts.i0, tt.i0), AT(tl.i1, ts.i0, tt.i1), AT(tl.i1, ts.i1, tt.i0), AT(tl.i1, ts.i1, tt.i1) };
#undef AT
    return blend8(t8, tt.w, ts.w, tl.w);
}

static f4 tex_lerp(const Ctx* c, float s)
{
    Tap t = make_tap(s / c->ref_step + 0.5f, c->n_ref);
    f4 a = c->ref_tex[t.i0], b = c->ref_tex[t.i1], r;
    if (g_rule.order >= 2) {
        const float w2[2] = {1.0f - t.w, t.w};
        const float vx[2] = {a.x, b.x}, vy[2] = {a.y, b.y}, vz[2] = {a.z, b.z}, vw[2] = {a.w, b.w};
        r.x = wsum(w2, vx, 2); r.y = wsum(w2, vy, 2); r.z = wsum(w2, vz, 2); r.w = wsum(w2, vw, 2);
    } else {
        r.x = lerp1(a.x, b.x, t.w); r.y = lerp1(a.y, b.y, t.w); r.z = lerp1(a.z, b.z, t.w); r.w = lerp1(a.w, b.w, t.w);
    }
    return r;
}

/* lat_lon_planner.cu:10-22 */
LLO_API void llo_dynamics(const TplLatLonState* x0, float dds, float dl, float dt, TplLatLonState* out)
{
    TplLatLonState res;
    memset(&res, 0, sizeof res);
    res.t = x0->t + dt;
    res.s = fmaxf(x0->s, x0->s + x0->ds * dt + 0.5 * dds * dt * dt);
    res.ds = fmaxf(0.0, x0->ds + dds * dt);
    res.dds = dds;
    res.l = x0->l + dl * dt;
    res.dl = dl;
    *out = res;
}

typedef struct {
    TplLatLonState tp;
    const Ctx* c;
    float dt, d_safety, d_fwd;
    int on_intersection;
    f4 rp;
} Evaluator;

static void ev_init(Evaluator* e, const TplLatLonState* tp, const Ctx* c, float dt)
{
    e->tp = *tp; e->c = c; e->dt = dt; e->d_safety = 0; e->d_fwd = 0;
    e->rp = tex_lerp(c, tp->s);
    e->on_intersection = roundf(e->rp.w) == 1.0f;
}

/* getMid, :83-124 -> (x, y, z) */
static void get_mid(const Evaluator* e, float* mx, float* my, float* mz)
{
    const TplLatLonParams* p = e->c->p;
    const float l_step_size = (p->l_max - p->l_min) / (float)(p->l_steps - 1);
    const float mean_dist = fmaxf(p->length_veh * 0.5, e->tp.ds * e->dt);
    float l_left = 0.0, l_right = 0.0;
    for (int i = 0; i < p->l_steps; ++i) {
        const float l = e->tp.l + i * l_step_size;
        const float d = dist_lon_x(e->c, e->tp.t, e->tp.s, l);
        if (d < mean_dist) { l_left = l; break; }
    }
    for (int i = 0; i < p->l_steps; ++i) {
        const float l = e->tp.l - i * l_step_size;
        const float d = dist_lon_x(e->c, e->tp.t, e->tp.s, l);
        if (d < mean_dist) { l_right = l; break; }
    }
    float mid = l_right + (l_left - l_right) * 0.5;
    *my = fminf(l_right + p->d_lat_comf, mid);
    *mz = fmaxf(l_left - p->d_lat_comf, mid);
    if (p->l_trg < l_right) *mx = *my;
    else if (p->l_trg > l_left) *mx = *mz;
    else *mx = p->l_trg;
}

static EvalResult final_state(const Evaluator* e)
{
    EvalResult res = {0.0f, 0.0f, 0};
    if (e->on_intersection) res.cost += e->c->p->w_xing_slow;
    float mx, my, mz;
    get_mid(e, &mx, &my, &mz);
    res.cost += e->c->p->w_l * sq(mx - e->tp.l);
    return res;
}

static EvalResult eval_state(const Evaluator* e)
{
    const TplLatLonParams* p = e->c->p;
    const TplLatLonState* tp = &e->tp;
    EvalResult res = {0.0f, 0.0f, 0};
    const float v_max_ref = e->rp.x;
    const float d_left_ref = (e->rp.y - p->width_veh * 0.5);
    const float d_right_ref = -(e->rp.z - p->width_veh * 0.5);
    res.cost += 1000.0 * fmaxf(0.0, tp->l - d_left_ref);
    res.cost += 1000.0 * fmaxf(0.0, d_right_ref - tp->l);
    float mx, my, mz;
    get_mid(e, &mx, &my, &mz);
    res.cost += p->w_l * sq(mx - tp->l);
    if (tp->l < my) res.cost += p->w_lat_dist * sq(my - tp->l);
    if (tp->l > mz) res.cost += p->w_lat_dist * sq(mz - tp->l);
    res.cost += p->w_progress * (1000.0 - tp->s);
    if (tp->ds > v_max_ref) { res.constr += tp->ds - v_max_ref; res.flags |= 2; }
    if (tp->t < p->t_st_min) res.cost += p->w_spatio_temporal * fmaxf(0.0, tp->s - p->s_st);
    if (tp->t > p->t_st_max) res.cost += p->w_spatio_temporal * fmaxf(0.0, p->s_st - tp->s);
    return res;
}

static EvalResult eval_action(const Evaluator* e, float dds, float dl, const TplLatLonState* tn)
{
    const TplLatLonParams* p = e->c->p;
    EvalResult res = {0.0f, 0.0f, 0};
    const float s_change = tn->s - e->tp.s;
    const float l_change = tn->l - e->tp.l;
    const float slope = fabsf(l_change / s_change);
    if (slope > p->slope_abs_max) { res.constr += fabsf(slope - p->slope_abs_max) * 1000.0; res.flags |= 4; }
    if (s_change > e->d_fwd) { res.constr += s_change - e->d_fwd; res.flags |= 1; }
    res.cost += p->w_safety_dist * fmaxf(0.0, s_change - e->d_safety);
    const float ddds = tn->dds - dds;
    const float ddl = tn->dl - dl;
    res.cost += p->w_dds * sq(dds * e->dt);
    res.cost += p->w_ddds * sq(ddds);
    res.cost += p->w_dl * sq(dl * e->dt);
    res.cost += p->w_ddl * sq(ddl);
    return res;
}

static int less_than(const EvalResult* a, const EvalResult* b)
{
    if (a->constr == b->constr) return a->cost < b->cost;
    return a->constr < b->constr;
}

/* next-stage fetch: table float[l][ds][s][4] */
static f4 fetch_point(const Ctx* c, const float* tab, float s, float ds, float l)
{
    const TplLatLonParams* p = c->p;
    float x = (s - p->s_min) / c->step_s, y = (ds - p->ds_min) / c->step_ds, z = (l - p->l_min) / c->step_l;
    int is = clampi((int)fminf(fmaxf(floorf(x + 0.5f), -1.0f), (float)p->s_steps), 0, p->s_steps - 1);
    int id = clampi((int)fminf(fmaxf(floorf(y + 0.5f), -1.0f), (float)p->ds_steps), 0, p->ds_steps - 1);
    int il = clampi((int)fminf(fmaxf(floorf(z + 0.5f), -1.0f), (float)p->l_steps), 0, p->l_steps - 1);
    const float* n = tab + 4 * (((size_t)il * p->ds_steps + id) * p->s_steps + is);
    f4 r = {n[0], n[1], n[2], n[3]};
    return r;
}

static f4 fetch_linear(const Ctx* c, const float* tab, float s, float ds, float l)
{
    const TplLatLonParams* p = c->p;
    Tap ts = make_tap((s - p->s_min) / c->step_s + 0.5f, p->s_steps);
    Tap td = make_tap((ds - p->ds_min) / c->step_ds + 0.5f, p->ds_steps);
    Tap tl = make_tap((l - p->l_min) / c->step_l + 0.5f, p->l_steps);
    f4 r;
    float* out = &r.x;
    for (int ch = 0; ch < 4; ++ch) {
#define AT(il, id, is) tab[4 * (((size_t)(il) * p->ds_steps + (id)) * p->s_steps + (is)) + ch]
        /* corners x + 2 y + 4 z with x = s, y = ds, z = l (default order: l, then ds, then s) */
        const float t8[8] = { AT(tl.i0, td.i0, ts.i0), AT(tl.i0, td.i0, ts.i1), AT(tl.i0, td.i1, ts.i0), AT(tl.i0, td.i1, ts.i1),
                              AT(tl.i1, td.i0, ts.i0), AT(tl.i1, td.i0, ts.i1), AT(tl.i1, td.i1, ts.i0), AT(tl.i1, td.i1, ts.i1) };
#undef AT
        out[ch] = blend8(t8, ts.w, td.w, tl.w);
    }
    return r;
}

/* findBestAction<SAMPLES_DDS, SAMPLES_DL, interp>, :206-273 */
static EvalResult find_best_action(Evaluator* e, int n_dds, int n_dl, int interp, float* dds_best, float* dl_best,
                                   EvalResult* min_res, const float* nodes_next)
{
    const TplLatLonParams* p = e->c->p;
    EvalResult total_min = {INFINITY, INFINITY, 0};
    const float dl_min = p->dl_min, dl_max = p->dl_max;
    const float step_size_dl = (dl_max - dl_min) / (float)(n_dl - 1);
    const int n_dl_2 = n_dl / 2;
    float d_fwd_left = 0.0, d_fwd_right = 0.0;
    for (int dl_idx = 0; dl_idx < n_dl; ++dl_idx) {
        float dl = 0.0;
        if (dl_idx == 0) {
            const float d = dist_lon_x(e->c, e->tp.t, e->tp.s, e->tp.l);
            d_fwd_left = d; d_fwd_right = d; e->d_fwd = d; dl = 0.0;
        } else if (dl_idx <= n_dl_2) {
            dl = step_size_dl * dl_idx;
            const float d = dist_lon_x(e->c, e->tp.t, e->tp.s, e->tp.l + dl * e->dt);
            d_fwd_left = fminf(d, d_fwd_left);
            e->d_fwd = d_fwd_left;
        } else {
            dl = -step_size_dl * (dl_idx - n_dl_2);
            const float d = dist_lon_x(e->c, e->tp.t, e->tp.s, e->tp.l + dl * e->dt);
            d_fwd_right = fminf(d, d_fwd_right);
            e->d_fwd = d_fwd_right;
        }
        e->d_fwd -= p->length_veh * 0.5;
        e->d_safety = e->d_fwd - p->gap_min - e->tp.ds * p->time_gap;
        for (int dds_idx = 0; dds_idx < n_dds; ++dds_idx) {
            const float dds = p->dds_min + (p->dds_max - p->dds_min) * ((float)dds_idx / (float)(n_dds - 1));
            TplLatLonState tn;
            llo_dynamics(&e->tp, dds, dl, e->dt, &tn);
            f4 nn = interp ? fetch_linear(e->c, nodes_next, tn.s, tn.ds, tn.l) : fetch_point(e->c, nodes_next, tn.s, tn.ds, tn.l);
            const float cost_next = nn.x, constr_next = nn.y;
            tn.dds = nn.z;
            tn.dl = nn.w;
            EvalResult res = eval_action(e, dds, dl, &tn);
            EvalResult total = {res.cost + cost_next, res.constr + constr_next, 0};
            if (less_than(&total, &total_min)) {
                total_min = total;
                *min_res = res;
                *dds_best = dds;
                *dl_best = dl;
            }
        }
    }
    return total_min;
}

static void ctx_init(Ctx* c, const TplLatLonParams* p, const TplDpEnvParams* ep, const float* dist_lon,
                     const double* ref_centered, int n_ref, float ref_step, f4* tex)
{
    const RefPt* rl = (const RefPt*)ref_centered;
    for (int i = 0; i < n_ref; ++i) {
        tex[i].x = rl[i].v_max; tex[i].y = rl[i].d_left; tex[i].z = rl[i].d_right; tex[i].w = rl[i].semantic;
    }
    c->p = p; c->ep = ep; c->dist_lon = dist_lon; c->ref_tex = tex; c->n_ref = n_ref; c->ref_step = ref_step;
    c->step_s = (p->s_max - p->s_min) / (float)(p->s_steps - 1);
    c->step_ds = (p->ds_max - p->ds_min) / (float)(p->ds_steps - 1);
    c->step_l = (p->l_max - p->l_min) / (float)(p->l_steps - 1);
}

/* updateTrajDp, :579-643.  tables: float[t][l][ds][s][4] (stage 0 unused); traj: LatLonState[t_steps],
 * traj[0] = start state. */
LLO_API void llo_update_traj_dp(const TplLatLonParams* p, const TplDpEnvParams* ep, const float* dist_lon,
                                const double* ref_centered, int n_ref, float ref_step,
                                float* tables, TplLatLonState* traj)
{
    f4* tex = (f4*)malloc(sizeof(f4) * (size_t)(n_ref > 0 ? n_ref : 1));
    Ctx c;
    ctx_init(&c, p, ep, dist_lon, ref_centered, n_ref, ref_step, tex);
    const size_t cells = (size_t)p->s_steps * p->ds_steps * p->l_steps;
    /* backward pass */
    for (int idx_t = p->t_steps - 1; idx_t > 0; --idx_t) {
        float* nodes = tables + 4 * cells * (size_t)idx_t;
        const int nx = idx_t + 1 < p->t_steps - 1 ? idx_t + 1 : p->t_steps - 1;
        const float* nodes_next = tables + 4 * cells * (size_t)nx;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
        for (int idx_l = 0; idx_l < p->l_steps; ++idx_l)
            for (int idx_ds = 0; idx_ds < p->ds_steps; ++idx_ds)
                for (int idx_s = 0; idx_s < p->s_steps; ++idx_s) {
                    TplLatLonState tp;
                    memset(&tp, 0, sizeof tp);
                    tp.t = p->dt_start + (idx_t - 1) * p->dt;
                    tp.s = p->s_min + idx_s * c.step_s;
                    tp.ds = p->ds_min + idx_ds * c.step_ds;
                    tp.l = p->l_min + idx_l * c.step_l;
                    Evaluator e;
                    ev_init(&e, &tp, &c, p->dt);
                    EvalResult res_state = eval_state(&e);
                    tp.cost = res_state.cost;
                    tp.constr = res_state.constr;
                    EvalResult total;
                    if (idx_t < p->t_steps - 1) {
                        EvalResult res_action = {0.0f, 0.0f, 0};
                        total = find_best_action(&e, 7, 7, 0, &tp.dds, &tp.dl, &res_action, nodes_next);
                    } else {
                        total = final_state(&e);
                    }
                    float* n = nodes + 4 * (((size_t)idx_l * p->ds_steps + idx_ds) * p->s_steps + idx_s);
                    n[0] = res_state.cost + total.cost;
                    n[1] = res_state.constr + total.constr;
                    n[2] = tp.dds;
                    n[3] = tp.dl;
                }
    }
    /* forward pass */
    for (int i = 0; i < p->t_steps; ++i) {
        const float dt = i == 0 ? p->dt_start : p->dt;
        const int nx = i + 1 < p->t_steps - 1 ? i + 1 : p->t_steps - 1;
        const float* nodes_next = tables + 4 * cells * (size_t)nx;
        TplLatLonState* tp = &traj[i];
        Evaluator e;
        ev_init(&e, tp, &c, dt);
        EvalResult res_state = eval_state(&e);
        tp->cost = res_state.cost;
        tp->constr = res_state.constr;
        tp->flags_constr = res_state.flags;
        if (i < p->t_steps - 1) {
            EvalResult res_action = {0.0f, 0.0f, 0};
            e.tp = *tp;
            find_best_action(&e, 21, 21, 1, &tp->dds, &tp->dl, &res_action, nodes_next);
            tp->cost += res_action.cost;
            tp->constr += res_action.constr;
            tp->flags_constr |= res_action.flags;
            llo_dynamics(tp, tp->dds, tp->dl, dt, &traj[i + 1]);
        }
    }
    free(tex);
}

/* reevalTraj, :357-404 + :819-859 */
LLO_API void llo_reeval_traj(const TplLatLonParams* p, const TplDpEnvParams* ep, const float* dist_lon,
                             const double* ref_centered, int n_ref, float ref_step, TplLatLonState* traj, int len)
{
    f4* tex = (f4*)malloc(sizeof(f4) * (size_t)(n_ref > 0 ? n_ref : 1));
    Ctx c;
    ctx_init(&c, p, ep, dist_lon, ref_centered, n_ref, ref_step, tex);
    for (int idx = 0; idx < len; ++idx) {
        TplLatLonState* tp = &traj[idx];
        float dt;
        if (idx == len - 1) dt = 0.0;
        else dt = traj[idx + 1].t - tp->t;
        Evaluator e;
        ev_init(&e, tp, &c, dt);
        EvalResult res_state = eval_state(&e);
        tp->cost = res_state.cost;
        tp->constr = res_state.constr;
        tp->flags_constr = res_state.flags;
        const float l_next = tp->l + dt * tp->dl;
        const float l_step_size = (p->l_max - p->l_min) / (float)(p->l_steps - 1);
        const float l_dist = l_next - tp->l;
        const int steps = ceilf(fabsf(l_dist) / l_step_size);
        const float step_size = l_dist / (float)(steps);
        float d_fwd = dist_lon_x(&c, tp->t, tp->s, tp->l);
        for (int i = 0; i < steps; ++i) {
            const float l = tp->l + i * step_size;
            const float d = dist_lon_x(&c, tp->t, tp->s, l);
            d_fwd = fminf(d, d_fwd);
        }
        e.tp = *tp;
        e.d_fwd = d_fwd;
        e.d_safety = d_fwd - p->gap_min - tp->ds * p->time_gap;
        if (idx < len - 1) {
            const TplLatLonState* tn = &traj[idx + 1];
            EvalResult res_action = eval_action(&e, tp->dds, tp->dl, tn);
            tp->cost += res_action.cost;
            tp->constr += res_action.constr;
            tp->flags_constr |= res_action.flags;
        }
    }
    free(tex);
}

/* ---- host side of the reference ---------------------------------------------------------- */

static void bracket(const TplLatLonState* v, int n, double t, int* i_prev, int* i_next, double* a)
{
    *i_prev = 0; *i_next = 0;
    if (t < v[0].t) { *i_prev = 0; *i_next = 1; }
    else if (t >= v[n - 1].t) { *i_prev = n - 2; *i_next = n - 1; }
    else {
        for (int i = 1; i < n; ++i) {
            const double t_prev = v[i - 1].t, t_next = v[i].t;
            if (t_prev <= t && t < t_next) { *i_prev = i - 1; *i_next = i; break; }
        }
    }
    const double t_prev = v[*i_prev].t, t_next = v[*i_next].t;
    *a = (t - t_prev) / (t_next - t_prev);
    *a = fmax(0.0, fmin(1.0, *a));
}

/* LatLonTraj::state, :432-441 */
LLO_API void llo_traj_state(const TplLatLonState* st, int n, float t, TplLatLonState* out)
{
    int ip, in; double a;
    bracket(st, n, t, &ip, &in, &a);
    float t_rel = t - st[ip].t;
    llo_dynamics(&st[ip], st[ip].dds, st[ip].dl, t_rel, out);
}

/* LatLonTraj::lerp, :443-463 */
LLO_API void llo_traj_lerp(const TplLatLonState* st, int n, float t, TplLatLonState* res)
{
    int ip, in; double a;
    bracket(st, n, t, &ip, &in, &a);
    float ai = 1.0f - a;
    memset(res, 0, sizeof *res);
    res->t = st[ip].t * ai + st[in].t * a;
    res->s = st[ip].s * ai + st[in].s * a;
    res->ds = st[ip].ds * ai + st[in].ds * a;
    res->dds = st[ip].dds * ai + st[in].dds * a;
    res->ddds = st[ip].ddds * ai + st[in].ddds * a;
    res->l = st[ip].l * ai + st[in].l * a;
    res->dl = st[ip].dl * ai + st[in].dl * a;
    res->ddl = st[ip].ddl * ai + st[in].ddl * a;
    res->dddl = st[ip].dddl * ai + st[in].dddl * a;
    res->cost = st[ip].cost * ai + st[in].cost * a;
    res->constr = st[ip].constr * ai + st[in].constr * a;
}

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

/* LatLonCartTraj::lerp, :465-486 */
LLO_API void llo_cart_lerp(const TplLatLonCartState* st, int n, float t, TplLatLonCartState* res)
{
    int ip = 0, in = 0;
    const double v = t;
    if (v < st[0].t) { ip = 0; in = 1; }
    else if (v >= st[n - 1].t) { ip = n - 2; in = n - 1; }
    else {
        for (int i = 1; i < n; ++i)
            if (st[i - 1].t <= v && v < st[i].t) { ip = i - 1; in = i; break; }
    }
    double a = (v - st[ip].t) / (st[in].t - st[ip].t);
    a = fmax(0.0, fmin(1.0, a));
    float ai = 1.0f - a;
    memset(res, 0, sizeof *res);
    res->t = st[ip].t * ai + st[in].t * a;
    res->distance = st[ip].distance * ai + st[in].distance * a;
    res->x = st[ip].x * ai + st[in].x * a;
    res->y = st[ip].y * ai + st[in].y * a;
    res->v = st[ip].v * ai + st[in].v * a;
    res->a = st[ip].a * ai + st[in].a * a;
    res->heading = st[ip].heading + short_angle_dist(st[ip].heading, st[in].heading) * a;
    res->k = st[ip].k * ai + st[in].k * a;
    res->cost = st[ip].cost * ai + st[in].cost * a;
    res->constr = st[ip].constr * ai + st[in].constr * a;
}

/* lqrSmoother<4, 1>, utils.hpp:278-348; matrices row major */
static void lqr4(const double* x0, const double* x_ref, int H, const double fx[16], const double fu[4],
                 const double lxx[16], double luu, double* xs)
{
    double* k = (double*)calloc((size_t)H, sizeof(double));
    double* K = (double*)calloc((size_t)H * 4, sizeof(double));
    double Vxx[16], Vx[4];
    memcpy(Vxx, lxx, sizeof Vxx);
    for (int i = 0; i < 4; ++i) {
        double acc = 0.0;
        for (int j = 0; j < 4; ++j) acc += lxx[4 * i + j] * x_ref[4 * (H - 1) + j];
        Vx[i] = -acc;
    }
    for (int it = H - 2; it >= 0; --it) {
        double lx[4], Qx[4], Qu = 0.0, Qxx[16], Quu, Qux[4], A[16], b[4];
        for (int i = 0; i < 4; ++i) {
            double acc = 0.0;
            for (int j = 0; j < 4; ++j) acc += lxx[4 * i + j] * x_ref[4 * it + j];
            lx[i] = -acc;
        }
        for (int i = 0; i < 4; ++i) {                       /* Qx = lx + fx^T Vx; Qu = fu^T Vx */
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += fx[4 * m + i] * Vx[m];
            Qx[i] = lx[i] + acc;
            Qu += fu[i] * Vx[i];
        }
        for (int i = 0; i < 4; ++i)                         /* A = fx^T Vxx */
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int m = 0; m < 4; ++m) acc += fx[4 * m + i] * Vxx[4 * m + j];
                A[4 * i + j] = acc;
            }
        for (int j = 0; j < 4; ++j) {                       /* b = fu^T Vxx */
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += fu[m] * Vxx[4 * m + j];
            b[j] = acc;
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int m = 0; m < 4; ++m) acc += A[4 * i + m] * fx[4 * m + j];
                Qxx[4 * i + j] = lxx[4 * i + j] + acc;
            }
        {
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += b[m] * fu[m];
            Quu = luu + acc;
        }
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += b[m] * fx[4 * m + j];
            Qux[j] = acc;
        }
        const double W = -1.0 / Quu;                         /* -Quu.inverse() */
        k[it] = W * Qu;
        for (int j = 0; j < 4; ++j) K[4 * it + j] = W * Qux[j];
        for (int i = 0; i < 4; ++i)
            Vx[i] = Qx[i] + K[4 * it + i] * Quu * k[it] + K[4 * it + i] * Qu + Qux[i] * k[it];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const double tmp_ij = K[4 * it + i] * Qux[j], tmp_ji = K[4 * it + j] * Qux[i];
                Vxx[4 * i + j] = Qxx[4 * i + j] + K[4 * it + i] * Quu * K[4 * it + j] + (tmp_ij + tmp_ji);
            }
    }
    for (int i = 0; i < 4; ++i) xs[i] = x0[i];
    for (int it = 0; it < H - 1; ++it) {
        double u = 0.0;
        for (int j = 0; j < 4; ++j) u += K[4 * it + j] * xs[4 * it + j];
        u += k[it];
        for (int i = 0; i < 4; ++i) {
            double acc = 0.0;
            for (int j = 0; j < 4; ++j) acc += fx[4 * i + j] * xs[4 * it + j];
            xs[4 * (it + 1) + i] = acc + fu[i] * u;
        }
    }
    free(k); free(K);
}

/* updateTrajSmooth, :645-760.  Returns the number of smoothed states written (<= cap). */
LLO_API int llo_update_traj_smooth(const TplLatLonParams* p, const TplLatLonState* traj_dp, int n_dp,
                                   const TplLatLonState* smooth0, TplLatLonState* out, int cap)
{
    int H = traj_dp[n_dp - 1].t / p->dt_smooth_traj;
    if (H < 2 || H > cap) return -1;
    double* ref_s = (double*)calloc((size_t)H * 4, sizeof(double));
    double* ref_l = (double*)calloc((size_t)H * 4, sizeof(double));
    double* xs_s = (double*)calloc((size_t)H * 4, sizeof(double));
    double* xs_l = (double*)calloc((size_t)H * 4, sizeof(double));
    for (int i = 0; i < H; ++i) {
        const float t = i * p->dt_smooth_traj;
        TplLatLonState tp;
        llo_traj_state(traj_dp, n_dp, t, &tp);
        ref_s[4 * i] = tp.s; ref_s[4 * i + 1] = tp.ds;
        ref_l[4 * i] = tp.l;
    }
    const double x0_s[4] = {smooth0->s, smooth0->ds, smooth0->dds, smooth0->ddds};
    const double x0_l[4] = {smooth0->l, smooth0->dl, smooth0->ddl, smooth0->dddl};
    double A[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    A[1] = p->dt_smooth_traj; A[6] = p->dt_smooth_traj; A[11] = p->dt_smooth_traj;
    const double B[4] = {0.0, 0.0, 0.0, p->dt_smooth_traj};
    double Qs[16] = {0}, Ql[16] = {0};
    Qs[0] = 10.0; Qs[5] = 10.0; Qs[10] = 10.0; Qs[15] = 10.0;
    Ql[0] = 1000.0; Ql[5] = 10.0; Ql[10] = 0.0; Ql[15] = 0.0;
    lqr4(x0_s, ref_s, H, A, B, Qs, 1.0, xs_s);
    lqr4(x0_l, ref_l, H, A, B, Ql, 0.1, xs_l);
    for (int i = 0; i < H; ++i) {
        TplLatLonState* tp = &out[i];
        memset(tp, 0, sizeof *tp);
        tp->t = i * p->dt_smooth_traj;
        tp->s = xs_s[4 * i]; tp->ds = xs_s[4 * i + 1]; tp->dds = xs_s[4 * i + 2]; tp->ddds = xs_s[4 * i + 3];
        tp->l = xs_l[4 * i]; tp->dl = xs_l[4 * i + 1]; tp->ddl = xs_l[4 * i + 2]; tp->dddl = xs_l[4 * i + 3];
    }
    free(ref_s); free(ref_l); free(xs_s); free(xs_l);
    return H;
}

/* interpRefLine, utils.hpp:148-173 */
static RefPt interp_ref_line(const RefPt* rl, int len, float step_size, float s)
{
    float a = s / step_size;
    int i0 = clampi((int)floorf(a), 0, len - 1);
    int i1 = clampi((int)ceilf(a), 0, len - 1);
    a -= i0;
    float ia = 1.0 - a;
    RefPt r;
    memset(&r, 0, sizeof r);
    r.x = rl[i0].x * ia + rl[i1].x * a;
    r.y = rl[i0].y * ia + rl[i1].y * a;
    r.heading = rl[i0].heading + short_angle_dist(rl[i0].heading, rl[i1].heading) * a;
    r.k = rl[i0].k * ia + rl[i1].k * a;
    return r;
}

/* frenetToCartesian, :768-817 */
LLO_API void llo_frenet_to_cartesian(const double* ref_centered, int n_ref, float ref_step, const double* offsets,
                                     const TplLatLonState* traj, int len_traj, TplLatLonCartState* result)
{
    const RefPt* rl = (const RefPt*)ref_centered;
    for (int i = 0; i < len_traj; ++i) {
        const TplLatLonState* tp = &traj[i];
        RefPt rp = interp_ref_line(rl, n_ref, ref_step, tp->s);
        TplLatLonCartState* cp = &result[i];
        memset(cp, 0, sizeof *cp);
        cp->t = tp->t;
        cp->x = offsets[0] + rp.x - (double)tp->l * sin(rp.heading);
        cp->y = offsets[1] + rp.y + (double)tp->l * cos(rp.heading);
        if (tp->ds < 1e-3) cp->heading = rp.heading;
        else cp->heading = atan((double)tp->dl / (double)tp->ds) + rp.heading;
        const double a = (1.0 - rp.k * tp->l) * tp->ds;
        cp->v = sqrt(a * a + (tp->dl * tp->dl));
        cp->cost = tp->cost;
        cp->constr = tp->constr;
        cp->flags_constr = tp->flags_constr;
    }
    for (int i = 1; i < len_traj; ++i) {
        TplLatLonCartState* cp0 = &result[i - 1];
        TplLatLonCartState* cp1 = &result[i];
        const float dx = cp1->x - cp0->x;
        const float dy = cp1->y - cp0->y;
        const float ds = sqrtf(dx * dx + dy * dy);
        cp1->distance = cp0->distance + ds;
        cp0->a = (cp1->v - cp0->v) / (cp1->t - cp0->t);
        if (ds >= 1e-3) cp0->k = short_angle_dist(cp0->heading, cp1->heading) / ds;
        else cp1->k = 0.0;
        cp1->k = cp0->k;
        cp1->a = cp0->a;
    }
}
