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
 *      promotes it); multiply-adds of one type are CONTRACTED the way nvcc's
 *      default -fmad=true contracts device code, written out as fma() calls
 *      (the sites are listed below); everything else is one IEEE-754 rounding
 *      per operation (build with -ffp-contract=off).
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
 * THE RULES ARE RUN-TIME STATE (round 6).  R1-R3 freeze choices the reference BINARY may make
 * differently (the rounding of the texture unit's 1.8 fixed-point weight and its internal blend
 * arithmetic are not documented; which multiply-adds nvcc fuses is its choice).  One shared object
 * serves every choice: dpo_set_rule(quant, lerp, fma, order) selects it for all later calls of this
 * library (process-wide; the OpenMP workers read it), dpo_get_rule reads it back, and the default is
 * what the reference BUILD most plausibly computes -- DpoRule{1, 0, 1, 0}:
 *   quant  R2  1 = rint(alpha*256)/256 (default), 2 = floor(alpha*256)/256, 0 = un-quantised weights
 *   lerp   R3  0 = fma(q, w, p*(1-w)) (default), 1 = p*(1-w) + q*w every operation rounded, 2 = p + w*(q-p)
 *   fma    R1  1 = CONTRACTED (default since round 6): nvcc compiles device code with -fmad=true unless
 *              told otherwise and the reference's build passes nothing that turns it off
 *              (library/CMakeLists.txt:89-99; its -ffast-math is forwarded to the HOST compiler).  The
 *              contraction is written out: every a*b + c of the reference source whose product and sum
 *              have the same type becomes fma(a, b, c), a*b + c*d becomes fma(a, b, c*d), a - b*c
 *              becomes fma(-b, c, a) -- the left-to-right rule of LLVM's DAG combiner, which is what
 *              NVVM applies.  A product whose sum is formed in another type (the float coordinate
 *              products that meet a double 0.5, lon_planner.cu:35; arr_tex_surf.cuh:148) has a
 *              conversion in between and stays unfused.  This is a MODEL of nvcc's fusion until it is
 *              checked against a dump or SASS.  The sites:
 *                lonDynamics  s: fmaf(v,dt,s) -> fma((.5a)dt, dt, .) -> fma(((j/6)dt)dt, dt, .)
 *                             v: fmaf(a,dt,v) -> fma((.5j)dt, dt, .);  a: fmaf(j,dt,a)
 *                interpPath   fmaf(p0, 1-alpha, p1*alpha) per field
 *                grid points  fmaf((float)i, step, min) for t, s, v, a and the dist-map stations
 *                evalNode     s_dist = fma(-len, 0.6, interp); the two double cost terms;
 *                             gap = fmaf(v, time_gap, gap_min) - s_dist; the action jerk
 *                             fmaf(j_max-j_min, k/(N-1), j_min); cost = fmaf(w_snap, d*d, cost)
 *              (sums whose other addend is the literal 0 -- "cost = 0; cost += w*x" -- give the
 *              product's own rounding either way.)
 *              0 = one IEEE rounding per operation of the source (the default until round 5).
 *              The file is built -ffp-contract=off: the compiler fuses nothing by itself.
 *   order  R3's ORDER / form.  0 = a -> v -> s (default: the order that lets the sweep kernels blend
 *              whole rows with row-uniform weights; nothing in the reference says so);
 *              1 = three sequential 1-D blends (the same lerp) along the station axis first (the
 *              texture's x), then velocity (y), then acceleration (z);
 *              2 = the symmetric form the CUDA programming guide documents for 3-D linear filtering:
 *              the eight corner weights w_c = (1-alpha|alpha)(1-beta|beta)(1-gamma|gamma) formed as
 *              float products ((w_s * w_v) * w_a: EXACT for 8-bit weights -- at most 24 significant
 *              bits), then sum_c w_c T_c accumulated in fp32 in the order the guide writes the terms
 *              (corner index = s + 2 v + 4 a), every product and every sum rounded, no fma; `lerp`
 *              does not apply;
 *              3 = the same sum with exact products and a double accumulator, rounded to float once.
 *              Both sums also decide the 2-D distance-map lookup (env.cu:260-263: x = t, y = s, z = 0;
 *              the t weight is 0 at the integral t index the planner asks for, so the sequential
 *              orders all give the blend along s alone -- lerp(p, 0-weight q) == p for finite q -- and
 *              only the sums differ).
 * The HIP planner carries the same four switches (tplhip_dp_lon_set_filter_rule / _set_contraction /
 * _set_blend_order) and is compared with this file under the same rule, bit for bit.
 */
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

/* The arithmetic rule every later call follows (see the header).  Process-wide on purpose: the checker is driven by one
 * test at a time, and the OpenMP workers of a sweep all read the same rule. */
typedef struct { int quant, lerp, fma, order; } DpoRule;
static DpoRule g_rule = { 1, 0, 1, 0 };

DPO_API int dpo_set_rule(int quant, int lerp, int fma, int order)
{
    if (quant < 0 || quant > 2 || lerp < 0 || lerp > 2 || fma < 0 || fma > 1 || order < 0 || order > 3) return -1;
    g_rule.quant = quant; g_rule.lerp = lerp; g_rule.fma = fma; g_rule.order = order;
    return 0;
}

DPO_API void dpo_get_rule(int out[4])
{
    out[0] = g_rule.quant; out[1] = g_rule.lerp; out[2] = g_rule.fma; out[3] = g_rule.order;
}

/* rule R1: a*b + c in float / in double, contracted or with its two roundings (the file is built -ffp-contract=off) */
static inline float  madf(float a, float b, float c, DpoRule r)    { return r.fma ? fmaf(a, b, c) : a * b + c; }
static inline double madd(double a, double b, double c, DpoRule r) { return r.fma ? fma(a, b, c) : a * b + c; }

/* grid point i of an axis (lon_planner.cu:196-199, env.cu:144): x_min + i * step under rule R1 */
DPO_API float dpo_grid_point(float x_min, float i, float step) { return madf(i, step, x_min, g_rule); }

/* ------------------------------------------------------------ dynamics -- */

/* lon_planner.cu:240-250 */
static inline void lon_dynamics_r(const TplLonState* x0, float j, float dt, TplLonState* out, DpoRule rl)
{
    TplLonState r;
    memset(&r, 0, sizeof r);            /* LonState members default to 0 (lon_planner.cuh:55-67) */
    r.t = x0->t + dt;
    {
        /* x0.s + x0.v*dt is float; the 0.5 and 1.0/6.0 literals promote the rest to double */
        float  sf = madf(x0->v, dt, x0->s, rl);
        double sd = madd((0.5 * (double)x0->a) * (double)dt, (double)dt, (double)sf, rl);
        sd = madd((((1.0 / 6.0) * (double)j) * (double)dt) * (double)dt, (double)dt, sd, rl);
        r.s = fmaxf(x0->s, (float)sd);
    }
    {
        float  vf = madf(x0->a, dt, x0->v, rl);
        double vd = madd((0.5 * (double)j) * (double)dt, (double)dt, (double)vf, rl);
        r.v = fmaxf(0.0f, (float)vd);
    }
    r.a = madf(j, dt, x0->a, rl);
    r.j = j;
    *out = r;
}

/* the DEVICE instantiation of the __interop__ function: follows rule R1 */
DPO_API void dpo_lon_dynamics(const TplLonState* x0, float j, float dt, TplLonState* out)
{
    lon_dynamics_r(x0, j, dt, out, g_rule);
}

/* the HOST instantiation (bindings_dp_lon.cu:85 `lon_dynamics`, LonTraj::state): compiled by the host compiler for
 * baseline x86-64, which has no fused multiply-add to contract into -- one rounding per operation whatever R1 says */
DPO_API void dpo_lon_dynamics_host(const TplLonState* x0, float j, float dt, TplLonState* out)
{
    DpoRule h = g_rule;
    h.fma = 0;
    lon_dynamics_r(x0, j, dt, out, h);
}

/* ---------------------------------------------------------------- path -- */

/* common.cuh:112-139; path is float[path_steps][7] = PathState */
static inline void interp_path_r(const float* path, float distance, float path_step_size,
                                 int path_steps, TplPathState* out, DpoRule rl)
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
        r[c] = madf(p0[c], alpha_inv, p1[c] * alpha, rl);
    }
    *out = res;
}

DPO_API void dpo_interp_path(const float* path, float distance, float path_step_size,
                             int path_steps, TplPathState* out)
{
    interp_path_r(path, distance, path_step_size, path_steps, out, g_rule);
}

/* ------------------------------------------------- texture filter (R2) -- */

typedef struct { int i0, i1; float w; } Tap;

static inline Tap make_tap(float x_plus_half, int n, DpoRule rl)
{
    Tap t;
    float xb = x_plus_half - 0.5f;
    float fl = floorf(xb);
    float alpha = xb - fl;
    if (rl.quant == 1)      t.w = rintf(alpha * 256.0f) / 256.0f;
    else if (rl.quant == 2) t.w = floorf(alpha * 256.0f) / 256.0f;
    else                    t.w = alpha;
    /* (int) of a float far outside int range is undefined; clamp in float first */
    float lo = fl < -1.0f ? -1.0f : (fl > (float)n ? (float)n : fl);
    int i = (int)lo;
    t.i0 = clampi(i, 0, n - 1);
    t.i1 = clampi(i + 1, 0, n - 1);
    return t;
}

/* R3, one 1-D blend (-ffp-contract=off keeps the roundings of forms 1 and 2) */
static inline float lerp1(float p, float q, float w, DpoRule rl)
{
    if (rl.lerp == 1) { const float a = p * (1.0f - w), b = q * w; return a + b; }
    if (rl.lerp == 2) { const float d = q - p; const float e = w * d; return p + e; }
    return fmaf(q, w, p * (1.0f - w));
}

/* sum_c w_c T_c over n corners in the order given: order 2 = fp32 accumulator, every product and sum rounded;
 * order 3 = exact products in a double accumulator with one final rounding */
static inline float wsum(const float* w, const float* t, int n, DpoRule rl)
{
    if (rl.order == 2) {
        float acc = w[0] * t[0];
        for (int c = 1; c < n; ++c) { const float pr = w[c] * t[c]; acc = acc + pr; }
        return acc;
    }
    double acc = (double)w[0] * (double)t[0];
    for (int c = 1; c < n; ++c) acc += (double)w[c] * (double)t[c];
    return (float)acc;
}

/* ------------------------------------------------------------ dist map -- */

static inline float env_t_idx(const TplDpEnvParams* ep, float t)
{
    /* env.cu:258: float t_idx = t < dt_start ? 0 : (roundf((t - dt_start) / dt) + 1.0); */
    if (t < ep->dt_start) return 0.0f;
    return (float)((double)roundf((t - ep->dt_start) / ep->dt) + 1.0);
}

/* env.cu:254-264 with get_cached (arr_tex_surf.cuh:142-149); dist is [t_steps][s_steps] */
static inline float interp_dist_map_path_r(const TplDpEnvParams* ep, const float* dist_ts,
                                           float t, float s, DpoRule rl)
{
    float t_idx = env_t_idx(ep, t);
    float sc = (s - ep->s_min) / (ep->s_max - ep->s_min) * (float)(ep->s_steps - 1);
    Tap tt = make_tap(t_idx + 0.5f, ep->t_steps, rl);
    Tap ts = make_tap(sc + 0.5f, ep->s_steps, rl);
    /* depth axis (size 1) always resolves to texel 0 with weight 0 */
    const float* r0 = dist_ts + (size_t)tt.i0 * ep->s_steps;
    const float* r1 = dist_ts + (size_t)tt.i1 * ep->s_steps;
    if (rl.order >= 2) {
        /* corner index = x + 2 y with x = t, y = s (the z weight is 0: those four terms are +0) */
        const float wt1 = tt.w, wt0 = 1.0f - tt.w, ws1 = ts.w, ws0 = 1.0f - ts.w;
        const float w[4] = { wt0 * ws0, wt1 * ws0, wt0 * ws1, wt1 * ws1 };
        const float t4[4] = { r0[ts.i0], r1[ts.i0], r0[ts.i1], r1[ts.i1] };
        return wsum(w, t4, 4, rl);
    }
    /* x (= t) first, then y (= s); with tt.w == 0 every sequential order gives the same value */
    float a = lerp1(r0[ts.i0], r1[ts.i0], tt.w, rl);      /* t_idx is integral -> tt.w == 0 */
    float b = lerp1(r0[ts.i1], r1[ts.i1], tt.w, rl);
    return lerp1(a, b, ts.w, rl);
}

DPO_API float dpo_interp_dist_map_path(const TplDpEnvParams* ep, const float* dist_ts,
                                        float t, float s)
{
    return interp_dist_map_path_r(ep, dist_ts, t, s, g_rule);
}

/* env.cu:131-158.  occ is [l_steps][s_steps][t_steps] (t fastest), point-read
 * at integer texel centres with clamp addressing. */
DPO_API void dpo_dist_map_path(const TplDpEnvParams* ep, const float* occ_lst,
                               const float* path, int path_steps, float path_step_size,
                               float* dist_ts)
{
    const DpoRule rl = g_rule;
    for (int idx_t = 0; idx_t < ep->t_steps; ++idx_t) {
        float d = 10000.0f;
        for (int idx_s = ep->s_steps - 1; idx_s >= 0; --idx_s) {
            const float s = madf((float)idx_s, ep->_s_step_size, ep->s_min, rl);
            TplPathState cps;
            interp_path_r(path, s, path_step_size, path_steps, &cps, rl);
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
static inline void read_node_r(const TplDpLonParams* p, const float* table,
                               float s, float v, float a, float out[4], DpoRule rl)
{
    float cs = (s - p->s_min) / (p->s_max - p->s_min) * (float)(p->s_steps - 1);
    float cv = (v - p->v_min) / (p->v_max - p->v_min) * (float)(p->v_steps - 1);
    float ca = (a - p->a_min) / (p->a_max - p->a_min) * (float)(p->a_steps - 1);
    /* "+ 0.5" is a double literal; tex3D takes float coordinates */
    Tap ts = make_tap((float)((double)cs + 0.5), p->s_steps, rl);
    Tap tv = make_tap((float)((double)cv + 0.5), p->v_steps, rl);
    Tap ta = make_tap((float)((double)ca + 0.5), p->a_steps, rl);

    const size_t S = (size_t)p->s_steps, V = (size_t)p->v_steps;
    const float* r00 = table + ((size_t)ta.i0 * V + tv.i0) * S * 4;   /* a0 v0 */
    const float* r10 = table + ((size_t)ta.i1 * V + tv.i0) * S * 4;   /* a1 v0 */
    const float* r01 = table + ((size_t)ta.i0 * V + tv.i1) * S * 4;   /* a0 v1 */
    const float* r11 = table + ((size_t)ta.i1 * V + tv.i1) * S * 4;   /* a1 v1 */
    out[3] = 0.0f;
    if (rl.order == 1) {
        for (int c = 0; c < 3; ++c) {                                 /* s, then v, then a */
            const size_t k0 = (size_t)ts.i0 * 4 + c, k1 = (size_t)ts.i1 * 4 + c;
            float x00 = lerp1(r00[k0], r00[k1], ts.w, rl);            /* a0 v0 */
            float x01 = lerp1(r01[k0], r01[k1], ts.w, rl);            /* a0 v1 */
            float x10 = lerp1(r10[k0], r10[k1], ts.w, rl);            /* a1 v0 */
            float x11 = lerp1(r11[k0], r11[k1], ts.w, rl);            /* a1 v1 */
            float y0 = lerp1(x00, x01, tv.w, rl);
            float y1 = lerp1(x10, x11, tv.w, rl);
            out[c] = lerp1(y0, y1, ta.w, rl);
        }
        return;
    }
    if (rl.order >= 2) {
        /* corner index = s + 2 v + 4 a, weights ((w_s * w_v) * w_a) */
        const float ws[2] = { 1.0f - ts.w, ts.w }, wv[2] = { 1.0f - tv.w, tv.w }, wa[2] = { 1.0f - ta.w, ta.w };
        const float* rows[4] = { r00, r01, r10, r11 };                 /* [2 a + v] */
        const size_t ks[2] = { (size_t)ts.i0 * 4, (size_t)ts.i1 * 4 };
        float w[8];
        for (int ia = 0; ia < 2; ++ia)
            for (int iv = 0; iv < 2; ++iv)
                for (int is = 0; is < 2; ++is) {
                    const float wsv = ws[is] * wv[iv];
                    w[is + 2 * iv + 4 * ia] = wsv * wa[ia];
                }
        for (int c = 0; c < 3; ++c) {
            float t8[8];
            for (int ia = 0; ia < 2; ++ia)
                for (int iv = 0; iv < 2; ++iv)
                    for (int is = 0; is < 2; ++is)
                        t8[is + 2 * iv + 4 * ia] = rows[2 * ia + iv][ks[is] + c];
            out[c] = wsum(w, t8, 8, rl);
        }
        return;
    }
    for (int c = 0; c < 3; ++c) {                                     /* a, then v, then s */
        float lo, hi;
        {
            size_t k = (size_t)ts.i0 * 4 + c;
            float b0 = lerp1(r00[k], r10[k], ta.w, rl);
            float b1 = lerp1(r01[k], r11[k], ta.w, rl);
            lo = lerp1(b0, b1, tv.w, rl);
        }
        {
            size_t k = (size_t)ts.i1 * 4 + c;
            float b0 = lerp1(r00[k], r10[k], ta.w, rl);
            float b1 = lerp1(r01[k], r11[k], ta.w, rl);
            hi = lerp1(b0, b1, tv.w, rl);
        }
        out[c] = lerp1(lo, hi, ts.w, rl);
    }
}

DPO_API void dpo_read_node(const TplDpLonParams* p, const float* table,
                           float s, float v, float a, float out[4])
{
    read_node_r(p, table, s, v, a, out, g_rule);
}

/* ------------------------------------------------------------- evalNode -- */

typedef struct { float cost, constr; } ActionResult;

/* lon_planner.cu:72-178 */
static void eval_node_r(const TplDpLonParams* p, const TplDpEnvParams* ep, int N,
                        TplLonState* tp, const float* table_next, const float* path,
                        const float* dist_ts, float dt, int choose_action,
                        int force_terminal, DpoRule rl)
{
    TplPathState cps;
    interp_path_r(path, tp->s, p->path_step_size, p->path_steps, &cps, rl);

    const float v_max = cps.v_max;
    const float s_dist = (float)madd(-(double)p->length_veh, 0.6,
                                     (double)interp_dist_map_path_r(ep, dist_ts, tp->t, cps.s, rl), rl);

    /* stateCost :40-55 */
    float state_cost = 0.0f;
    state_cost += p->w_a * (tp->a * tp->a);
    state_cost = (float)madd((double)p->w_progress, fabs(1000.0 - (double)tp->s), (double)state_cost, rl);
    {
        float gap = madf(tp->v, p->time_gap, p->gap_min, rl) - s_dist;
        double m = (double)gap > 0.0 ? (double)gap : 0.0;           /* max(0.0, gap) */
        state_cost = (float)madd((double)p->w_safety_dist, m, (double)state_cost, rl);
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
            ? madf(p->j_max - p->j_min, (float)j_idx / (float)(N - 1), p->j_min, rl)
            : tp->j;

        TplLonState tn;
        lon_dynamics_r(tp, j, dt, &tn, rl);

        float cost = state_cost;
        float constr = state_constr;

        float node[4];
        read_node_r(p, table_next, tn.s, tn.v, tn.a, node, rl);
        cost += node[0];
        constr += node[1];
        cost = madf(p->w_snap, (node[2] - j) * (node[2] - j), cost, rl);
        {
            float c = 0.0f;                                         /* stateActionCost :57-64 */
            c += p->w_j * ((j * dt) * (j * dt));
            cost += c;
        }

        TplPathState nps;
        interp_path_r(path, tn.s, p->path_step_size, p->path_steps, &nps, rl);
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
    tp->j = madf(p->j_max - p->j_min, (float)idx_min / (float)(N - 1), p->j_min, rl);
    tp->cost = cost_min;
    tp->constr = constr_min;
}

DPO_API void dpo_eval_node(const TplDpLonParams* p, const TplDpEnvParams* ep, int N,
                           TplLonState* tp, const float* table_next, const float* path,
                           const float* dist_ts, float dt, int choose_action,
                           int force_terminal)
{
    eval_node_r(p, ep, N, tp, table_next, path, dist_ts, dt, choose_action, force_terminal, g_rule);
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
    const DpoRule rl = g_rule;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for collapse(2) schedule(static) num_threads(nthreads)
#endif
    for (int ia = 0; ia < A; ++ia) {
        for (int iv = 0; iv < V; ++iv) {
            for (int is = 0; is < S; ++is) {
                TplLonState tp;
                memset(&tp, 0, sizeof tp);
                tp.t = madf((float)(idx_t - 1), p->dt, p->dt_start, rl);
                tp.s = madf((float)is, p->_s_step_size, p->s_min, rl);
                tp.v = madf((float)iv, p->_v_step_size, p->v_min, rl);
                tp.a = madf((float)ia, p->_a_step_size, p->a_min, rl);
                eval_node_r(p, ep, 9, &tp, table_next, path, dist_ts, p->dt, 1, force_terminal, rl);
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
    dpo_lon_dynamics_host(&states[i_prev], states[i_prev].j, t_rel, out);     /* LonTraj::state is host code */
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
