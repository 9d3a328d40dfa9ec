/*
 * dp_lon_oracle.c -- CPU restatement of tpl's DP (s,v,a) longitudinal planner.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tpl_amd/ may include, link or call
 * this file; it is the checker for the HIP path (tests/, __graft_entry__.smoke,
 * bench.py's cpu_baseline leg), never the thing shipped or measured.
 *
 * PARITY UNPINNED: the reference implementation is CUDA (3-D textures and
 * surfaces, OpenGL interop) and can be neither built nor run in this image, and
 * the reference's own tests hold no golden vector for this path.
 * This file restates the reference *source semantics* line by line:
 *
 *   lonDynamics            library/src/dyn_prog/lon_planner.cu:240-250
 *   interpPath             library/include/tplcpp/dyn_prog/common.cuh:112-139
 *   stateCost/ActionCost   lon_planner.cu:40-65
 *   evalNode<N>            lon_planner.cu:72-178
 *   lonBackwardStep        lon_planner.cu:180-209
 *   lonForwardStep         lon_planner.cu:211-225
 *   lonReevalNode          lon_planner.cu:227-236
 *   update / reevalTraj    lon_planner.cu:328-470 (driver order)
 *   updateDistMapPath      library/src/dyn_prog/env.cu:131-158
 *   interpDistMapPath      env.cu:254-264
 *   LonTraj::state         lon_planner.cu:252-260, utils.hpp:53-86
 *
 * Arithmetic rules frozen here (DESIGN.md "HP-1 arithmetic contract"):
 *  R1  Every expression is evaluated with the C++ types the reference source
 *      gives it (float, or double where a double literal / double variable
 *      promotes it), one IEEE-754 rounding per operation, NO fused
 *      multiply-add contraction (build with -ffp-contract=off), except where
 *      this file calls fmaf() explicitly (R3).
 *  R2  tex3D linear filtering ("note T"): unnormalised coordinate x (already
 *      including the reference's +0.5), xB = x - 0.5f, i = floor(xB),
 *      alpha = xB - i, alpha_q = rintf(alpha*256)/256 (8 fractional bits, the
 *      "9 bit fixed point" of arr_tex_surf.cuh:145), texel indices i and i+1
 *      clamped to [0, n-1] (cudaAddressModeClamp).
 *  R3  The trilinear blend order is: acceleration axis, then velocity axis,
 *      then station axis; each 1-D blend is  lerp(p,q,w) = fmaf(q, w, p*(1-w))
 *      with (1-w) exact.  All three stored channels (cost, constr, jerk) are
 *      blended.  The 1-D dist-map lookup uses the same lerp.
 *  R4  Stage t_steps-1 is always terminal in the backward sweep.  The
 *      reference binds that stage's own array as both surface and texture
 *      (lon_planner.cu:350-352), which is only well defined when the terminal
 *      test of :93 fires; this rule defines the remaining case.
 *
 * SENSITIVITY BUILDS.  R1-R3 freeze choices the reference BINARY may make
 * differently (nvcc contracts a*b+c into FMAs in device code by default; the
 * rounding of the texture unit's 1.8 fixed-point weight and its internal blend
 * arithmetic are not documented).  The default build is the contract above; the
 * switches below build variants of this file that tests/dp_sensitivity_study.py
 * runs against it to MEASURE how far those choices move tables and plans
 * ("HP-1 parity envelope").  They never build the checker the HIP
 * path is compared with.
 *   -DDPO_WEIGHT_QUANT=0   un-quantised filter weights
 *   -DDPO_WEIGHT_QUANT=2   weights truncated to 8 fractional bits (floor) instead of rint
 *   -DDPO_LERP_FORM=1      blend = p*(1-w) + q*w, every operation rounded (no fma)
 *   -DDPO_LERP_FORM=2      blend = p + w*(q-p)
 *   -ffp-contract=fast -mfma (instead of -ffp-contract=off): the compiler fuses
 *                          multiply-adds of lonDynamics / interpPath / the cost sums
 *                          like nvcc's default -fmad=true does -- but WHICH ones is
 *                          this compiler's choice (gcc), not nvcc's
 *   -DDPO_R1_FMA=1         (built -ffp-contract=off) the contraction written out: every
 *                          a*b + c of the reference source whose product and sum have the
 *                          same type becomes fma(a, b, c), a*b + c*d becomes
 *                          fma(a, b, c*d), a - b*c becomes fma(-b, c, a) -- the
 *                          left-to-right rule of LLVM's DAG combiner, which is what NVVM
 *                          applies under nvcc's default -fmad=true.  A product whose sum is
 *                          formed in another type (the float coordinate products that meet a
 *                          double 0.5, lon_planner.cu:35; arr_tex_surf.cuh:148) has a conversion
 *                          in between and stays unfused.  Deterministic across gcc / hipcc:
 *                          the HIP planner's tplhip_dp_lon_set_contraction(h, 1) is
 *                          compared with this build bit for bit.  The sites:
 *                            lonDynamics  s: fmaf(v,dt,s) -> fma((.5a)dt, dt, .) -> fma(((j/6)dt)dt, dt, .)
 *                                         v: fmaf(a,dt,v) -> fma((.5j)dt, dt, .);  a: fmaf(j,dt,a)
 *                            interpPath   fmaf(p0, 1-alpha, p1*alpha) per field
 *                            grid points  fmaf((float)i, step, min) for t, s, v, a and the dist-map stations
 *                            evalNode     s_dist = fma(-len, 0.6, interp); the two double cost terms;
 *                                         gap = fmaf(v, time_gap, gap_min) - s_dist; the action jerk
 *                                         fmaf(j_max-j_min, k/(N-1), j_min); cost = fmaf(w_snap, d*d, cost)
 *                          (sums whose other addend is the literal 0 -- "cost = 0; cost += w*x" --
 *                          give the product's own rounding either way.)
 *   -DDPO_BLEND_ORDER=1    rule R3's ORDER flipped: three sequential 1-D blends (the same lerp1) along the station axis
 *                          first (the texture's x), then velocity (y), then acceleration (z) -- what a sequential
 *                          implementation that starts with x would do.  The default a -> v -> s was chosen because it lets
 *                          the sweep kernels blend whole rows with row-uniform weights; nothing in the reference says so.
 *   -DDPO_BLEND_ORDER=2    the symmetric form the CUDA programming guide documents for 3-D linear filtering: the eight
 *                          corner weights w_c = (1-alpha|alpha)(1-beta|beta)(1-gamma|gamma) formed as float products
 *                          ((w_s * w_v) * w_a: EXACT for 8-bit weights -- at most 24 significant bits), then
 *                          sum_c w_c T_c accumulated in fp32 in the order the guide writes the terms (corner index =
 *                          s + 2 v + 4 a), every product and every sum rounded, no fma.  DPO_LERP_FORM does not apply.
 *   -DDPO_BLEND_ORDER=3    the same sum with exact products and a double accumulator, rounded to float once at the end
 *                          (a filter datapath wider than fp32).
 *                          Both sums also decide the 2-D distance-map lookup (env.cu:260-263: x = t, y = s, z = 0; the t
 *                          weight is 0 at the integral t index the planner asks for, so the sequential orders all give the
 *                          blend along s alone -- lerp1(p, 0-weight q) == p for finite q -- and only the sums differ).
 */
#ifndef DPO_WEIGHT_QUANT
#define DPO_WEIGHT_QUANT 1
#endif
#ifndef DPO_LERP_FORM
#define DPO_LERP_FORM 0
#endif
#ifndef DPO_R1_FMA
#define DPO_R1_FMA 0
#endif
#ifndef DPO_BLEND_ORDER
#define DPO_BLEND_ORDER 0
#endif
#include "../include/tplhip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define DPO_API __attribute__((visibility("default")))

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* rule R1: a*b + c in float / in double, unfused (default) or contracted (-DDPO_R1_FMA=1) */
#if DPO_R1_FMA
static inline float  madf(float a, float b, float c)    { return fmaf(a, b, c); }
static inline double madd(double a, double b, double c) { return fma(a, b, c); }
#else
static inline float  madf(float a, float b, float c)    { return a * b + c; }
static inline double madd(double a, double b, double c) { return a * b + c; }
#endif

/* ------------------------------------------------------------ dynamics -- */

/* lon_planner.cu:240-250 */
DPO_API void dpo_lon_dynamics(const TplLonState* x0, float j, float dt, TplLonState* out)
{
    TplLonState r;
    memset(&r, 0, sizeof r);            /* LonState members default to 0 (lon_planner.cuh:55-67) */
    r.t = x0->t + dt;
    {
        /* x0.s + x0.v*dt is float; the 0.5 and 1.0/6.0 literals promote the rest to double */
        float  sf = madf(x0->v, dt, x0->s);
        double sd = madd((0.5 * (double)x0->a) * (double)dt, (double)dt, (double)sf);
        sd = madd((((1.0 / 6.0) * (double)j) * (double)dt) * (double)dt, (double)dt, sd);
        r.s = fmaxf(x0->s, (float)sd);
    }
    {
        float  vf = madf(x0->a, dt, x0->v);
        double vd = madd((0.5 * (double)j) * (double)dt, (double)dt, (double)vf);
        r.v = fmaxf(0.0f, (float)vd);
    }
    r.a = madf(j, dt, x0->a);
    r.j = j;
    *out = r;
}

/* ---------------------------------------------------------------- path -- */

/* common.cuh:112-139; path is float[path_steps][7] = PathState */
DPO_API void dpo_interp_path(const float* path, float distance, float path_step_size,
                             int path_steps, TplPathState* out)
{
    TplPathState res;
    memset(&res, 0, sizeof res);
    if (path == NULL) { *out = res; return; }

    float alpha = distance / path_step_size;
    /* max(0, min(n-1, (int)floor(alpha))); the float->int conversion saturates
     * on the device, which the float-side guard reproduces (plain C would be
     * undefined for |alpha| >= 2^31) */
    float flo = floorf(alpha), fhi = ceilf(alpha);
    flo = flo < -1.0f ? -1.0f : (flo > (float)path_steps ? (float)path_steps : flo);
    fhi = fhi < -1.0f ? -1.0f : (fhi > (float)path_steps ? (float)path_steps : fhi);
    int i0 = clampi((int)flo, 0, path_steps - 1);
    int i1 = clampi((int)fhi, 0, path_steps - 1);
    alpha -= (float)i0;
    float alpha_inv = (float)(1.0 - (double)alpha);

    const float* p0 = path + (size_t)i0 * 7;
    const float* p1 = path + (size_t)i1 * 7;
    float* r = (float*)&res;
    for (int c = 0; c < 7; ++c) {
        r[c] = madf(p0[c], alpha_inv, p1[c] * alpha);
    }
    *out = res;
}

/* ------------------------------------------------- texture filter (R2) -- */

typedef struct { int i0, i1; float w; } Tap;

static inline Tap make_tap(float x_plus_half, int n)
{
    Tap t;
    float xb = x_plus_half - 0.5f;
    float fl = floorf(xb);
    float alpha = xb - fl;
#if DPO_WEIGHT_QUANT == 1
    t.w = rintf(alpha * 256.0f) / 256.0f;
#elif DPO_WEIGHT_QUANT == 2
    t.w = floorf(alpha * 256.0f) / 256.0f;
#else
    t.w = alpha;
#endif
    /* (int) of a float far outside int range is undefined; clamp in float first */
    float lo = fl < -1.0f ? -1.0f : (fl > (float)n ? (float)n : fl);
    int i = (int)lo;
    t.i0 = clampi(i, 0, n - 1);
    t.i1 = clampi(i + 1, 0, n - 1);
    return t;
}

static inline float lerp1(float p, float q, float w)
{
#if DPO_LERP_FORM == 0
    return fmaf(q, w, p * (1.0f - w));
#elif DPO_LERP_FORM == 1
    volatile float a = p * (1.0f - w), b = q * w;       /* volatile: keeps the two products rounded in every build */
    return a + b;
#else
    volatile float d = q - p;
    volatile float e = w * d;
    return p + e;
#endif
}

#if DPO_BLEND_ORDER >= 2
/* sum_c w_c T_c over n corners in the order given: fp32 accumulator (every product and sum rounded; volatile keeps
 * the roundings in every build) or exact products in a double accumulator with one final rounding */
static inline float wsum(const float* w, const float* t, int n)
{
#if DPO_BLEND_ORDER == 2
    volatile float acc = w[0] * t[0];
    for (int c = 1; c < n; ++c) { volatile float pr = w[c] * t[c]; acc = acc + pr; }
    return acc;
#else
    double acc = (double)w[0] * (double)t[0];
    for (int c = 1; c < n; ++c) acc += (double)w[c] * (double)t[c];
    return (float)acc;
#endif
}
#endif

/* ------------------------------------------------------------ dist map -- */

static inline float env_t_idx(const TplDpEnvParams* ep, float t)
{
    /* env.cu:258: float t_idx = t < dt_start ? 0 : (roundf((t - dt_start) / dt) + 1.0); */
    if (t < ep->dt_start) return 0.0f;
    return (float)((double)roundf((t - ep->dt_start) / ep->dt) + 1.0);
}

/* env.cu:254-264 with get_cached (arr_tex_surf.cuh:142-149); dist is [t_steps][s_steps] */
DPO_API float dpo_interp_dist_map_path(const TplDpEnvParams* ep, const float* dist_ts,
                                        float t, float s)
{
    float t_idx = env_t_idx(ep, t);
    float sc = (s - ep->s_min) / (ep->s_max - ep->s_min) * (float)(ep->s_steps - 1);
    Tap tt = make_tap(t_idx + 0.5f, ep->t_steps);
    Tap ts = make_tap(sc + 0.5f, ep->s_steps);
    /* depth axis (size 1) always resolves to texel 0 with weight 0 */
    const float* r0 = dist_ts + (size_t)tt.i0 * ep->s_steps;
    const float* r1 = dist_ts + (size_t)tt.i1 * ep->s_steps;
#if DPO_BLEND_ORDER >= 2
    /* corner index = x + 2 y with x = t, y = s (the z weight is 0: those four terms are +0) */
    const float wt1 = tt.w, wt0 = 1.0f - tt.w, ws1 = ts.w, ws0 = 1.0f - ts.w;
    const float w[4] = { wt0 * ws0, wt1 * ws0, wt0 * ws1, wt1 * ws1 };
    const float t4[4] = { r0[ts.i0], r1[ts.i0], r0[ts.i1], r1[ts.i1] };
    return wsum(w, t4, 4);
#else
    /* x (= t) first, then y (= s); with tt.w == 0 every sequential order gives the same value */
    float a = lerp1(r0[ts.i0], r1[ts.i0], tt.w);      /* t_idx is integral -> tt.w == 0 */
    float b = lerp1(r0[ts.i1], r1[ts.i1], tt.w);
    return lerp1(a, b, ts.w);
#endif
}

/* env.cu:131-158.  occ is [l_steps][s_steps][t_steps] (t fastest), point-read
 * at integer texel centres with clamp addressing. */
DPO_API void dpo_dist_map_path(const TplDpEnvParams* ep, const float* occ_lst,
                               const float* path, int path_steps, float path_step_size,
                               float* dist_ts)
{
    for (int idx_t = 0; idx_t < ep->t_steps; ++idx_t) {
        float d = 10000.0f;
        for (int idx_s = ep->s_steps - 1; idx_s >= 0; --idx_s) {
            const float s = madf((float)idx_s, ep->_s_step_size, ep->s_min);
            TplPathState cps;
            dpo_interp_path(path, s, path_step_size, path_steps, &cps);
            float fs = (cps.s - ep->s_min) / ep->_s_step_size;
            float fl = (cps.l - ep->l_min) / ep->_l_step_size;
            /* (int) truncation; guard the undefined out-of-range conversion */
            fs = fs < -1.0f ? -1.0f : (fs > (float)ep->s_steps ? (float)ep->s_steps : fs);
            fl = fl < -1.0f ? -1.0f : (fl > (float)ep->l_steps ? (float)ep->l_steps : fl);
            int is = clampi((int)fs, 0, ep->s_steps - 1);
            int il = clampi((int)fl, 0, ep->l_steps - 1);
            float occ = occ_lst[((size_t)il * ep->s_steps + is) * ep->t_steps + idx_t];
            if (occ > 0.0f) {
                d = 0.0f;
            } else {
                d += ep->_s_step_size;
            }
            dist_ts[(size_t)idx_t * ep->s_steps + idx_s] = d;
        }
    }
}

/* ---------------------------------------------------------- value table -- */

/* readNode lon_planner.cu:26-38; table is float[a][v][s][4] */
DPO_API void dpo_read_node(const TplDpLonParams* p, const float* table,
                           float s, float v, float a, float out[4])
{
    float cs = (s - p->s_min) / (p->s_max - p->s_min) * (float)(p->s_steps - 1);
    float cv = (v - p->v_min) / (p->v_max - p->v_min) * (float)(p->v_steps - 1);
    float ca = (a - p->a_min) / (p->a_max - p->a_min) * (float)(p->a_steps - 1);
    /* "+ 0.5" is a double literal; tex3D takes float coordinates */
    Tap ts = make_tap((float)((double)cs + 0.5), p->s_steps);
    Tap tv = make_tap((float)((double)cv + 0.5), p->v_steps);
    Tap ta = make_tap((float)((double)ca + 0.5), p->a_steps);

    const size_t S = (size_t)p->s_steps, V = (size_t)p->v_steps;
    const float* r00 = table + ((size_t)ta.i0 * V + tv.i0) * S * 4;   /* a0 v0 */
    const float* r10 = table + ((size_t)ta.i1 * V + tv.i0) * S * 4;   /* a1 v0 */
    const float* r01 = table + ((size_t)ta.i0 * V + tv.i1) * S * 4;   /* a0 v1 */
    const float* r11 = table + ((size_t)ta.i1 * V + tv.i1) * S * 4;   /* a1 v1 */
#if DPO_BLEND_ORDER == 1
    for (int c = 0; c < 3; ++c) {                                     /* s, then v, then a */
        const size_t k0 = (size_t)ts.i0 * 4 + c, k1 = (size_t)ts.i1 * 4 + c;
        float x00 = lerp1(r00[k0], r00[k1], ts.w);                    /* a0 v0 */
        float x01 = lerp1(r01[k0], r01[k1], ts.w);                    /* a0 v1 */
        float x10 = lerp1(r10[k0], r10[k1], ts.w);                    /* a1 v0 */
        float x11 = lerp1(r11[k0], r11[k1], ts.w);                    /* a1 v1 */
        float y0 = lerp1(x00, x01, tv.w);
        float y1 = lerp1(x10, x11, tv.w);
        out[c] = lerp1(y0, y1, ta.w);
    }
    out[3] = 0.0f;
    return;
#elif DPO_BLEND_ORDER >= 2
    {
        /* corner index = s + 2 v + 4 a, weights ((w_s * w_v) * w_a) */
        const float ws[2] = { 1.0f - ts.w, ts.w }, wv[2] = { 1.0f - tv.w, tv.w }, wa[2] = { 1.0f - ta.w, ta.w };
        const float* rows[4] = { r00, r01, r10, r11 };                 /* [2 a + v] */
        const size_t ks[2] = { (size_t)ts.i0 * 4, (size_t)ts.i1 * 4 };
        float w[8];
        for (int ia = 0; ia < 2; ++ia)
            for (int iv = 0; iv < 2; ++iv)
                for (int is = 0; is < 2; ++is) {
                    volatile float wsv = ws[is] * wv[iv];
                    w[is + 2 * iv + 4 * ia] = wsv * wa[ia];
                }
        for (int c = 0; c < 3; ++c) {
            float t8[8];
            for (int ia = 0; ia < 2; ++ia)
                for (int iv = 0; iv < 2; ++iv)
                    for (int is = 0; is < 2; ++is)
                        t8[is + 2 * iv + 4 * ia] = rows[2 * ia + iv][ks[is] + c];
            out[c] = wsum(w, t8, 8);
        }
        out[3] = 0.0f;
        return;
    }
#endif
    for (int c = 0; c < 3; ++c) {
        float lo, hi;
        {
            size_t k = (size_t)ts.i0 * 4 + c;
            float b0 = lerp1(r00[k], r10[k], ta.w);
            float b1 = lerp1(r01[k], r11[k], ta.w);
            lo = lerp1(b0, b1, tv.w);
        }
        {
            size_t k = (size_t)ts.i1 * 4 + c;
            float b0 = lerp1(r00[k], r10[k], ta.w);
            float b1 = lerp1(r01[k], r11[k], ta.w);
            hi = lerp1(b0, b1, tv.w);
        }
        out[c] = lerp1(lo, hi, ts.w);
    }
    out[3] = 0.0f;
}

/* ------------------------------------------------------------- evalNode -- */

typedef struct { float cost, constr; } ActionResult;

/* lon_planner.cu:72-178 */
DPO_API void dpo_eval_node(const TplDpLonParams* p, const TplDpEnvParams* ep, int N,
                           TplLonState* tp, const float* table_next, const float* path,
                           const float* dist_ts, float dt, int choose_action,
                           int force_terminal)
{
    TplPathState cps;
    dpo_interp_path(path, tp->s, p->path_step_size, p->path_steps, &cps);

    const float v_max = cps.v_max;
    const float s_dist = (float)madd(-(double)p->length_veh, 0.6,
                                     (double)dpo_interp_dist_map_path(ep, dist_ts, tp->t, cps.s));

    /* stateCost :40-55 */
    float state_cost = 0.0f;
    state_cost += p->w_a * (tp->a * tp->a);
    state_cost = (float)madd((double)p->w_progress, fabs(1000.0 - (double)tp->s), (double)state_cost);
    {
        float gap = madf(tp->v, p->time_gap, p->gap_min) - s_dist;
        double m = (double)gap > 0.0 ? (double)gap : 0.0;           /* max(0.0, gap) */
        state_cost = (float)madd((double)p->w_safety_dist, m, (double)state_cost);
    }

    float state_constr = 0.0f;
    if (tp->v > v_max) {
        state_constr += tp->v - v_max;
    }

    if (force_terminal || roundf(tp->t / p->dt) == (float)(p->t_steps - 1)) {
        tp->cost = state_cost;
        return;
    }

    ActionResult results[64];
    if (N > 64) N = 64;

    const int n_eval = choose_action ? N : 1;
    for (int j_idx = 0; j_idx < n_eval; ++j_idx) {
        const float j = choose_action
            ? madf(p->j_max - p->j_min, (float)j_idx / (float)(N - 1), p->j_min)
            : tp->j;

        TplLonState tn;
        dpo_lon_dynamics(tp, j, dt, &tn);

        float cost = state_cost;
        float constr = state_constr;

        float node[4];
        dpo_read_node(p, table_next, tn.s, tn.v, tn.a, node);
        cost += node[0];
        constr += node[1];
        cost = madf(p->w_snap, (node[2] - j) * (node[2] - j), cost);
        {
            float c = 0.0f;                                         /* stateActionCost :57-64 */
            c += p->w_j * ((j * dt) * (j * dt));
            cost += c;
        }

        TplPathState nps;
        dpo_interp_path(path, tn.s, p->path_step_size, p->path_steps, &nps);
        const double v_max_n = (double)nps.v_max;
        if ((double)tn.v > v_max_n) {
            constr = (float)((double)constr + ((double)tn.v - v_max_n));
        }

        const float s_change = tn.s - tp->s;
        if (s_change > s_dist) {
            constr += s_change - s_dist;
        }
        if (tn.a < p->a_min) {
            constr += p->a_min - tn.a;
        }
        if (tn.a > p->a_max) {
            constr += tn.a - p->a_max;
        }
        results[j_idx].cost = cost;
        results[j_idx].constr = constr;
    }

    if (!choose_action) {
        tp->cost = results[0].cost;
        tp->constr = results[0].constr;
        return;
    }

    float constr_min = INFINITY;
    for (int k = 0; k < N; ++k) {
        if (results[k].constr < constr_min) constr_min = results[k].constr;
    }
    float cost_min = INFINITY;
    int idx_min = 0;
    for (int k = 0; k < N; ++k) {
        if (results[k].constr > constr_min) continue;
        if (results[k].cost < cost_min) {
            cost_min = results[k].cost;
            idx_min = k;
        }
    }
    tp->j = madf(p->j_max - p->j_min, (float)idx_min / (float)(N - 1), p->j_min);
    tp->cost = cost_min;
    tp->constr = constr_min;
}

/* --------------------------------------------------------------- sweeps -- */

/* lonBackwardStep lon_planner.cu:180-209 over the whole grid of stage idx_t.
 * table_out / table_next: float[a][v][s][4].  nthreads <= 0: all cores. */
DPO_API void dpo_backward_stage(const TplDpLonParams* p, const TplDpEnvParams* ep, int idx_t,
                                const float* table_next, float* table_out,
                                const float* path, const float* dist_ts, int nthreads)
{
    const int S = p->s_steps, V = p->v_steps, A = p->a_steps;
    const int force_terminal = (idx_t == p->t_steps - 1);           /* rule R4 */
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for collapse(2) schedule(static) num_threads(nthreads)
#endif
    for (int ia = 0; ia < A; ++ia) {
        for (int iv = 0; iv < V; ++iv) {
            for (int is = 0; is < S; ++is) {
                TplLonState tp;
                memset(&tp, 0, sizeof tp);
                tp.t = madf((float)(idx_t - 1), p->dt, p->dt_start);
                tp.s = madf((float)is, p->_s_step_size, p->s_min);
                tp.v = madf((float)iv, p->_v_step_size, p->v_min);
                tp.a = madf((float)ia, p->_a_step_size, p->a_min);
                dpo_eval_node(p, ep, 9, &tp, table_next, path, dist_ts, p->dt, 1, force_terminal);
                float* o = table_out + (((size_t)ia * V + iv) * S + is) * 4;
                o[0] = tp.cost; o[1] = tp.constr; o[2] = tp.j; o[3] = 0.0f;
            }
        }
    }
    (void)nthreads;
}

DPO_API size_t dpo_table_floats(const TplDpLonParams* p)
{
    return (size_t)p->s_steps * p->v_steps * p->a_steps * 4;
}

/* DynProgLonPlanner::update lon_planner.cu:328-401.
 * tables: float[t_steps][a][v][s][4] scratch/out (stage 0 is never written, as
 * in the reference); dist_ts: [env t_steps][env s_steps], already computed.
 * inits/out_trajs: n rollouts sharing the backward sweep. */
DPO_API int dpo_update(const TplDpLonParams* p, const TplDpEnvParams* ep,
                       const float* path, const float* dist_ts,
                       int n, const TplLonState* inits,
                       float* tables, TplLonState* out_trajs, int nthreads)
{
    const int T = p->t_steps;
    const size_t tf = dpo_table_floats(p);
    if (T < 1) return -1;

    for (int i = T - 1; i > 0; --i) {
        int nx = i + 1 < T - 1 ? i + 1 : T - 1;
        dpo_backward_stage(p, ep, i, tables + (size_t)nx * tf, tables + (size_t)i * tf,
                           path, dist_ts, nthreads);
    }
    for (int r = 0; r < n; ++r) {
        TplLonState* traj = out_trajs + (size_t)r * T;
        traj[0] = inits[r];
        for (int i = 0; i < T; ++i) {
            float dt = (i == 0) ? p->dt_start : p->dt;
            int nx = i + 1 < T - 1 ? i + 1 : T - 1;
            dpo_eval_node(p, ep, 21, &traj[i], tables + (size_t)nx * tf, path, dist_ts, dt, 1, 0);
            if (i < T - 1) {
                dpo_lon_dynamics(&traj[i], traj[i].j, dt, &traj[i + 1]);
            }
        }
    }
    return 0;
}

/* DynProgLonPlanner::reevalTraj lon_planner.cu:403-470 */
DPO_API int dpo_reeval(const TplDpLonParams* p, const TplDpEnvParams* ep,
                       const float* path, const float* dist_ts,
                       const float* tables, const TplLonState* traj, int len,
                       TplLonState* out)
{
    const int T = p->t_steps;
    const size_t tf = dpo_table_floats(p);
    if (len != T) return -3;
    memcpy(out, traj, sizeof(TplLonState) * (size_t)len);
    for (int i = 0; i < len; ++i) {
        float dt = (i == 0) ? p->dt_start : p->dt;
        int nx = i + 1 < T - 1 ? i + 1 : T - 1;
        dpo_eval_node(p, ep, 21, &out[i], tables + (size_t)nx * tf, path, dist_ts, dt, 0, 0);
    }
    return 0;
}

/* LonTraj::state lon_planner.cu:252-260; InterpVars utils.hpp:53-86 */
DPO_API int dpo_traj_state(const TplLonState* states, int n, float t, TplLonState* out)
{
    if (n < 2) return -1;
    const double v = (double)t;
    int i_prev = 0;
    if (v < (double)states[0].t) {
        i_prev = 0;
    } else if (v >= (double)states[n - 1].t) {
        i_prev = n - 2;
    } else {
        for (int i = 1; i < n; ++i) {
            if ((double)states[i - 1].t <= v && v < (double)states[i].t) {
                i_prev = i - 1;
                break;
            }
        }
    }
    float t_rel = t - states[i_prev].t;
    dpo_lon_dynamics(&states[i_prev], states[i_prev].j, t_rel, out);
    return 0;
}

DPO_API void dpo_update_step_sizes(TplDpLonParams* p)
{
    p->_s_step_size = (p->s_max - p->s_min) / (float)(p->s_steps - 1);
    p->_v_step_size = (p->v_max - p->v_min) / (float)(p->v_steps - 1);
    p->_a_step_size = (p->a_max - p->a_min) / (float)(p->a_steps - 1);
}

DPO_API void dpo_env_update_step_sizes(TplDpEnvParams* p)
{
    p->_l_step_size = (p->l_max - p->l_min) / (float)(p->l_steps - 1);
    p->_s_step_size = (p->s_max - p->s_min) / (float)(p->s_steps - 1);
}

DPO_API int dpo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
