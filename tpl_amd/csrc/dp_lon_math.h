// dp_lon_math.h -- device-side arithmetic of the DP (s,v,a) planner for gfx950.
//
// Every function states the reference lines it follows.  The arithmetic
// contract (DESIGN.md "HP-1 arithmetic contract", rules R1-R4) is:
//   R1  types exactly as the reference source promotes them; multiply-adds of one
//       type contracted the way nvcc's default -fmad=true contracts device code,
//       written out as fma() (madf / madd below); everything else one rounding per
//       operation (this TU is built -ffp-contract=off: the compiler fuses nothing);
//   R2  tex3D linear filter = floor/frac of (x-0.5), weight quantised to 8
//       fractional bits, clamp addressing;
//   R3  trilinear order a -> v -> s, lerp(p,q,w) = fmaf(q, w, p*(1-w));
//   R4  stage t_steps-1 is always terminal in the backward sweep.
#pragma once

#include <hip/hip_runtime.h>
#include "../../include/tplhip.h"

namespace tplhip {

struct Tap { int i0, i1; float w; };

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

// R2 and R3 are what the reference's TEXTURE UNIT does (arr_tex_surf.cuh:142-149) -- undocumented arithmetic that
// this repository could only freeze, not observe.  For HP-1 they are therefore parameters:
// the general kernels of dp_lon.hip take the rule as an argument (tplhip_dp_lon_set_filter_rule), so that a dump of
// the CUDA build that disagrees with the frozen choice is answered by a call, not by a rewrite.  Everything else in the
// library (and the plain-fp32 sweeps of dp_sweep.hip) is built for the default rule: a default argument folds away.
//   quant: 1 = weight rint(a * 256) / 256 (default), 2 = floor(a * 256) / 256, 0 = not quantised
//   lerp : 0 = fma(q, w, p * (1 - w)) (default), 1 = p * (1 - w) + q * w with every operation rounded, 2 = p + w * (q - p)
//   fma  : rule R1.  1 = the contraction nvcc's default -fmad=true applies to the reference's device code (DEFAULT since
//          round 6: the reference's build passes nothing that turns it off, library/CMakeLists.txt:89-99), written out:
//          a*b + c -> fma(a, b, c), a*b + c*d -> fma(a, b, c*d), a - b*c -> fma(-b, c, a) wherever product and sum have the
//          same type (madf / madd below; the sites are listed in the header of oracle/dp_lon_oracle.c) -- a MODEL of nvcc's
//          fusion until checked against a dump;  0 = one rounding per operation of the reference source (the default until
//          round 5).  Set by tplhip_dp_lon_set_contraction; a run-time branch in the general kernels, a template bit of the
//          plain-fp32 sweeps (kFma in dp_sweep.hip) -- both settings are built with every division form.
//   order: rule R3's ORDER / form (round 5; tplhip_dp_lon_set_blend_order).  0 = three sequential blends a -> v -> s (default:
//          the order that lets the sweeps blend whole rows with row-uniform weights), 1 = s -> v -> a (the texture's x first),
//          2 = the symmetric eight-corner sum  sum_c w_c T_c  with w_c = (w_s * w_v) * w_a (exact for 8-bit weights), products and
//          sums rounded in fp32 in the order of the documented formula (corner = s + 2 v + 4 a), 3 = the same sum with exact
//          products in a double accumulator and one rounding.  `lerp` does not apply to 2 and 3.  The distance map's 2-D
//          lookup (x = t with weight 0, y = s) is the same for every sequential order and becomes a four-corner sum under 2 / 3.
//          oracle/dp_lon_oracle.c under dpo_set_rule(., ., ., n) is the same arithmetic.
struct FilterRule { int quant, lerp, fma, order; };
constexpr FilterRule kDefaultRule{1, 0, 1, 0};
__host__ __device__ constexpr bool same_rule(FilterRule a, FilterRule b) { return a.quant == b.quant && a.lerp == b.lerp && a.fma == b.fma && a.order == b.order; }

// a*b + c under rule R1 (this file is built -ffp-contract=off: the unfused form keeps its two roundings)
__device__ __forceinline__ float madf(float a, float b, float c, FilterRule r) { return r.fma ? fmaf(a, b, c) : a * b + c; }
__device__ __forceinline__ double madd(double a, double b, double c, FilterRule r) { return r.fma ? fma(a, b, c) : a * b + c; }
// grid point i of an axis, lon_planner.cu:196-199 / env.cu:144
__device__ __forceinline__ float grid_pt(float x_min, float i, float step, FilterRule r = kDefaultRule) { return madf(i, step, x_min, r); }

__device__ __forceinline__ float quant_weight(float alpha, FilterRule r)
{
    if (r.quant == 1) return rintf(alpha * 256.0f) / 256.0f;          // power-of-two scaling: exact
    if (r.quant == 2) return floorf(alpha * 256.0f) / 256.0f;
    return alpha;
}

// R2.  x_plus_half is the coordinate as handed to tex3D (arr_tex_surf.cuh:142-149).
__device__ __forceinline__ Tap make_tap(float x_plus_half, int n, FilterRule rule = kDefaultRule)
{
    Tap t;
    float xb = x_plus_half - 0.5f;
    float fl = floorf(xb);
    float alpha = xb - fl;
    t.w = quant_weight(alpha, rule);
    float lo = fminf(fmaxf(fl, -1.0f), (float)n);
    int i = (int)lo;
    t.i0 = clampi(i, 0, n - 1);
    t.i1 = clampi(i + 1, 0, n - 1);
    return t;
}

// R3 (this translation unit is built -ffp-contract=off: the forms below keep their roundings)
__device__ __forceinline__ float lerp1(float p, float q, float w, FilterRule rule = kDefaultRule)
{
    if (rule.lerp == 1) { const float a = p * (1.0f - w), b = q * w; return a + b; }
    if (rule.lerp == 2) { const float d = q - p; const float e = w * d; return p + e; }
    return fmaf(q, w, p * (1.0f - w));
}

__device__ __forceinline__ float4 lerp4(float4 p, float4 q, float w, FilterRule rule = kDefaultRule)
{
    return make_float4(lerp1(p.x, q.x, w, rule), lerp1(p.y, q.y, w, rule), lerp1(p.z, q.z, w, rule), 0.0f);
}

// sum_c w[c] t[c] over N corners in index order (rule.order 2: fp32, every product and sum rounded -- the file is built
// -ffp-contract=off; 3: exact products, double accumulator, one rounding)
template <int N>
__device__ __forceinline__ float wsum(const float (&w)[N], const float (&t)[N], FilterRule rule)
{
    if (rule.order == 2) {
        float acc = w[0] * t[0];
#pragma unroll
        for (int c = 1; c < N; ++c) { const float pr = w[c] * t[c]; acc = acc + pr; }
        return acc;
    }
    double acc = (double)w[0] * (double)t[0];
#pragma unroll
    for (int c = 1; c < N; ++c) acc += (double)w[c] * (double)t[c];
    return (float)acc;
}

// corner weights of the eight-corner sum, index s + 2 v + 4 a
__device__ __forceinline__ void corner_weights(float ws, float wv, float wa, float (&w)[8])
{
    const float s2[2] = {1.0f - ws, ws}, v2[2] = {1.0f - wv, wv}, a2[2] = {1.0f - wa, wa};
#pragma unroll
    for (int c = 0; c < 8; ++c) { const float sv = s2[c & 1] * v2[(c >> 1) & 1]; w[c] = sv * a2[c >> 2]; }
}

// R3 over the eight corners of one channel, t[s + 2 v + 4 a]
__device__ __forceinline__ float blend8(const float (&t)[8], float ws, float wv, float wa, FilterRule rule)
{
    if (rule.order == 1) {                                      // s, then v, then a
        const float x00 = lerp1(t[0], t[1], ws, rule), x01 = lerp1(t[2], t[3], ws, rule);      // a0: v0, v1
        const float x10 = lerp1(t[4], t[5], ws, rule), x11 = lerp1(t[6], t[7], ws, rule);      // a1: v0, v1
        return lerp1(lerp1(x00, x01, wv, rule), lerp1(x10, x11, wv, rule), wa, rule);
    }
    if (rule.order >= 2) {
        float w[8];
        corner_weights(ws, wv, wa, w);
        return wsum<8>(w, t, rule);
    }
    const float lo = lerp1(lerp1(t[0], t[4], wa, rule), lerp1(t[2], t[6], wa, rule), wv, rule);  // a, then v, then s
    const float hi = lerp1(lerp1(t[1], t[5], wa, rule), lerp1(t[3], t[7], wa, rule), wv, rule);
    return lerp1(lo, hi, ws, rule);
}

// the same for the three stored channels of eight nodes n[s + 2 v + 4 a]
__device__ __forceinline__ float4 blend8_nodes(const float4 (&n)[8], float ws, float wv, float wa, FilterRule rule)
{
    float c[8], g[8], j[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c[i] = n[i].x; g[i] = n[i].y; j[i] = n[i].z; }
    if (rule.order >= 2) {
        float w[8];
        corner_weights(ws, wv, wa, w);
        return make_float4(wsum<8>(w, c, rule), wsum<8>(w, g, rule), wsum<8>(w, j, rule), 0.0f);
    }
    return make_float4(blend8(c, ws, wv, wa, rule), blend8(g, ws, wv, wa, rule), blend8(j, ws, wv, wa, rule), 0.0f);
}

// The 2-D distance-map blend (env.cu:260-263: x = t, y = s): corners (t0,s0) (t1,s0) (t0,s1) (t1,s1).  Sequential orders
// blend along t (x) first -- with the integral t index the planner asks for, wt == 0 and every order gives the blend along
// s alone; the sums (order 2 / 3) take the four corners in the documented order x + 2 y.
__device__ __forceinline__ float dist_blend(float p00, float p10, float p01, float p11, float wt, float ws, FilterRule rule)
{
    if (rule.order >= 2) {
        const float wt0 = 1.0f - wt, ws0 = 1.0f - ws;
        const float w[4] = {wt0 * ws0, wt * ws0, wt0 * ws, wt * ws};
        const float t[4] = {p00, p10, p01, p11};
        return wsum<4>(w, t, rule);
    }
    return lerp1(lerp1(p00, p10, wt, rule), lerp1(p01, p11, wt, rule), ws, rule);
}

// One cell of a value table in HBM: 12 bytes.  The reference keeps float4 texels whose fourth
// component is never read (lon_planner.cu:34-37); dropping it cuts a quarter of the bytes every
// sweep pulls through L2 -- which is what bounds the sweep.  In registers and LDS a node travels
// as a float4 with w = 0.
struct Node { float c, g, j; };
static_assert(sizeof(Node) == 12, "Node must be 12 bytes");

__device__ __forceinline__ float4 ld_node(const Node* __restrict__ p)
{
    const Node n = *p;
    return make_float4(n.c, n.g, n.j, 0.0f);
}

__device__ __forceinline__ void st_node(Node* __restrict__ p, float c, float g, float j)
{
    Node n; n.c = c; n.g = g; n.j = j;
    *p = n;
}

// 12-byte buffer load: descriptor + 32-bit lane offset + scalar offset
typedef float v3f32 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float4 buf_load_node(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset)
{
    const v3f32 r = __builtin_bit_cast(v3f32, __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)voffset, (int)soffset, 0));
    return make_float4(r.x, r.y, r.z, 0.0f);
}

// lerp4 with the complement 1-w supplied (wave-uniform weights keep it in an SGPR)
__device__ __forceinline__ float4 lerp4w(float4 p, float4 q, float w, float iw)
{
    float4 r;
    r.x = fmaf(q.x, w, p.x * iw);
    r.y = fmaf(q.y, w, p.y * iw);
    r.z = fmaf(q.z, w, p.z * iw);
    r.w = 0.0f;
    return r;
}

// lonDynamics, lon_planner.cu:240-250
__device__ __forceinline__ TplLonState lon_dynamics(const TplLonState& x0, float j, float dt, FilterRule rule = kDefaultRule)
{
    TplLonState r;
    r.t = x0.t + dt;
    float sf = madf(x0.v, dt, x0.s, rule);
    double sd = madd((0.5 * (double)x0.a) * (double)dt, (double)dt, (double)sf, rule);
    sd = madd((((1.0 / 6.0) * (double)j) * (double)dt) * (double)dt, (double)dt, sd, rule);
    r.s = fmaxf(x0.s, (float)sd);
    float vf = madf(x0.a, dt, x0.v, rule);
    double vd = madd((0.5 * (double)j) * (double)dt, (double)dt, (double)vf, rule);
    r.v = fmaxf(0.0f, (float)vd);
    r.a = madf(j, dt, x0.a, rule);
    r.j = j;
    r.cost = 0.0f;
    r.constr = 0.0f;
    return r;
}

// interpPath, common.cuh:112-139, for one column of the path.  col points at
// element 0 of that column, stride in floats.
// (float)(1.0 - (double)alpha) == 1.0f - alpha for every float alpha: the
// double difference is exact unless |alpha| < 2^-29, where both round to 1.0f.
struct PathTap { int i0, i1; float a, ia; };

__device__ __forceinline__ PathTap path_tap(float distance, float path_step_size, int path_steps)
{
    PathTap t;
    float alpha = distance / path_step_size;
    // guard the float->int conversion of far out-of-range values (UB in C)
    float lo = fminf(fmaxf(floorf(alpha), -1.0f), (float)path_steps);
    float hi = fminf(fmaxf(ceilf(alpha), -1.0f), (float)path_steps);
    t.i0 = clampi((int)lo, 0, path_steps - 1);
    t.i1 = clampi((int)hi, 0, path_steps - 1);
    alpha -= (float)t.i0;
    t.a = alpha;
    t.ia = 1.0f - alpha;
    return t;
}

__device__ __forceinline__ float path_lerp(const PathTap& t, float p0, float p1, FilterRule rule = kDefaultRule)
{
    return madf(p0, t.ia, p1 * t.a, rule);
}

// path_tap with the quotient distance / path_step_size already formed.
// i0 = clamp((int)floor(a), 0, n-1), i1 = clamp((int)ceil(a), 0, n-1) (common.cuh:124-125), with
// ceil(a) = floor(a) + (a > floor(a)).  The clamps are done on the (integral) floats -- exact
// below 2^24 -- so (float)i0 needs no conversion back and each clamp is one v_med3_f32.
__device__ __forceinline__ PathTap path_tap_q(float alpha, int path_steps)
{
    PathTap t;
    const float top = (float)(path_steps - 1);
    const float fl = floorf(alpha);
    const float f0 = __builtin_amdgcn_fmed3f(fl, 0.0f, top);
    const float f1 = __builtin_amdgcn_fmed3f(fl + (alpha > fl ? 1.0f : 0.0f), 0.0f, top);
    t.i0 = (int)f0;
    t.i1 = (int)f1;
    alpha -= f0;
    t.a = alpha;
    t.ia = 1.0f - alpha;
    return t;
}

// env.cu:258
__device__ __forceinline__ float env_t_idx(const TplDpEnvParams& ep, float t)
{
    if (t < ep.dt_start) return 0.0f;
    return (float)((double)roundf((t - ep.dt_start) / ep.dt) + 1.0);
}

// interpDistMapPath, env.cu:254-264; dist is [t_steps][s_steps]
template <typename PtrT>
__device__ __forceinline__ float interp_dist_map_path(const TplDpEnvParams& ep, PtrT dist_ts,
                                                      float t, float s, FilterRule rule = kDefaultRule)
{
    float t_idx = env_t_idx(ep, t);
    float sc = (s - ep.s_min) / (ep.s_max - ep.s_min) * (float)(ep.s_steps - 1);
    Tap tt = make_tap(t_idx + 0.5f, ep.t_steps, rule);
    Tap ts = make_tap(sc + 0.5f, ep.s_steps, rule);
    PtrT r0 = dist_ts + (size_t)tt.i0 * ep.s_steps;
    PtrT r1 = dist_ts + (size_t)tt.i1 * ep.s_steps;
    return dist_blend(r0[ts.i0], r1[ts.i0], r0[ts.i1], r1[ts.i1], tt.w, ts.w, rule);
}

// coordinate of readNode, lon_planner.cu:34-37: float expression + double 0.5 -> float.
// (float)((double)c + 0.5) == c + 0.5f for every float c: the double sum is exact, so both
// are the single rounding of the exact sum to float.
__device__ __forceinline__ float node_coord(float x, float x_min, float x_max, int steps)
{
    float c = (x - x_min) / (x_max - x_min) * (float)(steps - 1);
    return c + 0.5f;
}

// readNode with all eight taps from global memory (forward / reeval kernels).
__device__ __forceinline__ float4 read_node(const TplDpLonParams& p, const Node* __restrict__ table,
                                            float s, float v, float a, FilterRule rule = kDefaultRule)
{
    Tap ts = make_tap(node_coord(s, p.s_min, p.s_max, p.s_steps), p.s_steps, rule);
    Tap tv = make_tap(node_coord(v, p.v_min, p.v_max, p.v_steps), p.v_steps, rule);
    Tap ta = make_tap(node_coord(a, p.a_min, p.a_max, p.a_steps), p.a_steps, rule);
    const size_t S = (size_t)p.s_steps, V = (size_t)p.v_steps;
    const Node* r00 = table + ((size_t)ta.i0 * V + tv.i0) * S;
    const Node* r10 = table + ((size_t)ta.i1 * V + tv.i0) * S;
    const Node* r01 = table + ((size_t)ta.i0 * V + tv.i1) * S;
    const Node* r11 = table + ((size_t)ta.i1 * V + tv.i1) * S;
    if (rule.order != 0) {                                      // another blend order / the eight-corner sums
        const float4 n[8] = {ld_node(r00 + ts.i0), ld_node(r00 + ts.i1), ld_node(r01 + ts.i0), ld_node(r01 + ts.i1),
                             ld_node(r10 + ts.i0), ld_node(r10 + ts.i1), ld_node(r11 + ts.i0), ld_node(r11 + ts.i1)};
        return blend8_nodes(n, ts.w, tv.w, ta.w, rule);
    }
    float4 lo = lerp4(lerp4(ld_node(r00 + ts.i0), ld_node(r10 + ts.i0), ta.w, rule),
                      lerp4(ld_node(r01 + ts.i0), ld_node(r11 + ts.i0), ta.w, rule), tv.w, rule);
    float4 hi = lerp4(lerp4(ld_node(r00 + ts.i1), ld_node(r10 + ts.i1), ta.w, rule),
                      lerp4(ld_node(r01 + ts.i1), ld_node(r11 + ts.i1), ta.w, rule), tv.w, rule);
    return lerp4(lo, hi, ts.w, rule);
}

// stateCost, lon_planner.cu:40-55
__device__ __forceinline__ float state_cost(const TplDpLonParams& p, float s, float v, float a, float s_dist, FilterRule rule = kDefaultRule)
{
    float cost = 0.0f;
    cost += p.w_a * (a * a);
    cost = (float)madd((double)p.w_progress, fabs(1000.0 - (double)s), (double)cost, rule);
    float gap = madf(v, p.time_gap, p.gap_min, rule) - s_dist;
    double m = fmax(0.0, (double)gap);
    cost = (float)madd((double)p.w_safety_dist, m, (double)cost, rule);
    return cost;
}

// s_dist of evalNode, lon_planner.cu:84: distance-map value minus 0.6 vehicle lengths, formed in double
__device__ __forceinline__ float s_dist_of(float dist_value, const TplDpLonParams& p, FilterRule rule = kDefaultRule)
{
    return (float)madd(-(double)p.length_veh, 0.6, (double)dist_value, rule);
}

// jerk of action k out of N, lon_planner.cu:148-149
__device__ __forceinline__ float action_jerk(const TplDpLonParams& p, int k, int N, FilterRule rule = kDefaultRule)
{
    return madf(p.j_max - p.j_min, (float)k / (float)(N - 1), p.j_min, rule);
}

struct ActionResult { float cost, constr; };

// evalAction lambda, lon_planner.cu:99-137, with the node value already looked up
__device__ __forceinline__ ActionResult eval_action_tail(const TplDpLonParams& p,
                                                          float state_c, float state_g,
                                                          float4 node, float j, float wj_cost,
                                                          float tn_s, float tn_v, float tn_a,
                                                          float tp_s, float s_dist, float v_max_next, FilterRule rule = kDefaultRule)
{
    float cost = state_c;
    float constr = state_g;
    cost += node.x;
    constr += node.y;
    cost = madf(p.w_snap, (node.z - j) * (node.z - j), cost, rule);
    cost += wj_cost;
    const double v_max = (double)v_max_next;
    if ((double)tn_v > v_max) {
        constr = (float)((double)constr + ((double)tn_v - v_max));
    }
    const float s_change = tn_s - tp_s;
    if (s_change > s_dist) {
        constr += s_change - s_dist;
    }
    if (tn_a < p.a_min) {
        constr += p.a_min - tn_a;
    }
    if (tn_a > p.a_max) {
        constr += tn_a - p.a_max;
    }
    ActionResult r;
    r.cost = cost;
    r.constr = constr;
    return r;
}

// stateActionCost, lon_planner.cu:57-64
__device__ __forceinline__ float state_action_cost(const TplDpLonParams& p, float j, float dt)
{
    float c = 0.0f;
    c += p.w_j * ((j * dt) * (j * dt));
    return c;
}

} // namespace tplhip
