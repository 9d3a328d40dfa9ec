// lat_lon.hip -- SURVEY 8(f-3): the DP (s, ds, l) lateral/longitudinal planner for gfx950.
// Replaces library/src/dyn_prog/lat_lon_planner.cu behind the reference's DynProgLatLonPlanner
// surface (bindings_dp_lat_lon.cu).
//
// Re-design, not a translation:
//  * value tables are plain HBM arrays float4[t][l][ds][s] (s fastest); the backward sweep's
//    point-sampled texture fetch is an index computation (texel = floor(coordinate + 0.5),
//    clamped), the forward pass' filtered fetch the software filter of rules R2/R3;
//  * everything a cell needs from the environment depends on (stage, station) only -- the
//    reference-line texel (speed limit, lane bounds, intersection flag) and the profile of the
//    longitudinal distance map across the environment's lateral cells.  k_ll_prelude computes
//    both ONCE per (stage, station); the reference re-filters the distance map up to
//    2*l_steps + 7 times in every one of the s*ds*l cells of a stage;
//  * the stages of the backward pass are enqueued back to back, and the forward pass -- one
//    <<<1,1>>> launch per stage in the reference, each scanning 441 actions serially -- is ONE
//    launch of one wavefront with the 441 actions across its lanes and a lexicographic wave
//    argmin that keeps the reference's first-minimum-in-loop-order tie-break;
//  * trajectory smoothing (two 4-state LQR passes) and the Frenet->Cartesian conversion are host
//    code in the reference and here.
#include <hip/hip_runtime.h>

#include <array>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "dp_env_internal.h"
#include "dp_lon_math.h"

using namespace tplhip;

namespace {

constexpr unsigned CONSTR_OCCUPANCY = 1, CONSTR_VELOCITY = 2, CONSTR_ANGLE = 4;   // lat_lon_planner.cuh:105-109

struct Eval { float cost, constr; unsigned flags; };

// EvalResult::operator<, lat_lon_planner.cu:36-43
__host__ __device__ inline bool eval_less(float cost_a, float constr_a, float cost_b, float constr_b)
{
    if (constr_a == constr_b) return cost_a < cost_b;
    return constr_a < constr_b;
}

// dynamics, lat_lon_planner.cu:10-22
__host__ __device__ inline TplLatLonState ll_dynamics(const TplLatLonState& x0, float dds, float dl, float dt)
{
    TplLatLonState r;
    r.t = x0.t + dt;
    const double sd = (double)(x0.s + x0.ds * dt) + ((0.5 * (double)dds) * (double)dt) * (double)dt;
    r.s = fmaxf(x0.s, (float)sd);
    r.ds = fmaxf(0.0f, x0.ds + dds * dt);
    r.dds = dds;
    r.ddds = 0.0f;
    r.l = x0.l + dl * dt;
    r.dl = dl;
    r.ddl = 0.0f; r.dddl = 0.0f; r.cost = 0.0f; r.constr = 0.0f; r.flags_constr = 0;
    return r;
}

__host__ __device__ inline float sqf(float x) { return x * x; }

// interpDistMapLon(t, s, l).x, env.cu:236-252; dist_lon is float[l][s][t][2]
__device__ __forceinline__ float dist_lon_x(const float* __restrict__ dist_lon, const TplDpEnvParams& ep,
                                            float t, float s, float l, FilterRule rule = kDefaultRule)
{
    const float t_idx = env_t_idx(ep, t);
    const float sc = (s - ep.s_min) / (ep.s_max - ep.s_min) * (float)(ep.s_steps - 1);
    const float lc = roundf((l - ep.l_min) / (ep.l_max - ep.l_min) * (float)(ep.l_steps - 1));
    const Tap tt = make_tap(t_idx + 0.5f, ep.t_steps, rule);
    const Tap ts = make_tap(sc + 0.5f, ep.s_steps, rule);
    const Tap tl = make_tap(lc + 0.5f, ep.l_steps, rule);
    auto at = [&](int il, int is, int it) { return dist_lon[(((size_t)il * ep.s_steps + is) * ep.t_steps + it) * 2]; };
    // corners x + 2 y + 4 z of the texture: x = t, y = s, z = l (the default order blends l, then s, then t)
    const float t8[8] = {at(tl.i0, ts.i0, tt.i0), at(tl.i0, ts.i0, tt.i1), at(tl.i0, ts.i1, tt.i0), at(tl.i0, ts.i1, tt.i1),
                         at(tl.i1, ts.i0, tt.i0), at(tl.i1, ts.i0, tt.i1), at(tl.i1, ts.i1, tt.i0), at(tl.i1, ts.i1, tt.i1)};
    return blend8(t8, tt.w, ts.w, tl.w, rule);
}

// a 1-D filtered fetch under the rule: the two-corner sum for the sum forms (order 2 / 3), the blend otherwise
__device__ __forceinline__ float lerp_1d(float a, float b, float w, FilterRule rule)
{
    if (rule.order >= 2) { const float w2[2] = {1.0f - w, w}; const float t2[2] = {a, b}; return wsum<2>(w2, t2, rule); }
    return lerp1(a, b, w, rule);
}

// ref_line.texLerp(s), common.cuh:93-95: (v_max, d_left, d_right, semantic) of the float texture
__device__ __forceinline__ float4 ref_tex(const float4* __restrict__ ref, int n_ref, float ref_step, float s, FilterRule rule = kDefaultRule)
{
    const Tap tp = make_tap(s / ref_step + 0.5f, n_ref, rule);
    const float4 a = ref[tp.i0], b = ref[tp.i1];
    return make_float4(lerp_1d(a.x, b.x, tp.w, rule), lerp_1d(a.y, b.y, tp.w, rule), lerp_1d(a.z, b.z, tp.w, rule), lerp_1d(a.w, b.w, tp.w, rule));
}

// Source of interpDistMapLon(tp.t, tp.s, .) for one (t, s): either the precomputed profile across
// the environment's lateral cells (backward sweep) or the distance map itself.
struct DistSrc {
    const float* profile;            // [env l_steps] or nullptr
    const float* dist_lon;
    float t, s;
    FilterRule rule = kDefaultRule;  // the texture filter's rules (tplhip_lat_lon_set_filter_rule)
};

__device__ __forceinline__ float dist_at(const DistSrc& d, const TplDpEnvParams& ep, float l)
{
    if (d.profile) {
        const float lc = roundf((l - ep.l_min) / (ep.l_max - ep.l_min) * (float)(ep.l_steps - 1));
        // integral coordinate: weight 0, texel i0 -- under every rule (the profile was filtered along s and t with the rule; a weight of
        // exactly 0 along l leaves x for the finite distances of the map in the blends and adds +0 terms in the sums)
        const Tap tl = make_tap(lc + 0.5f, ep.l_steps, d.rule);
        return d.profile[tl.i0];
    }
    return dist_lon_x(d.dist_lon, ep, d.t, d.s, l, d.rule);
}

// Evaluator::getMid, lat_lon_planner.cu:83-124.  WAVE_PAR: called by a whole, converged wavefront
// with identical arguments (the forward rollout); the outward walks of both sides then run one
// offset per lane -- lanes 0..31 to the left, 32..63 to the right -- and the first hit in the
// reference's order is the lowest set bit of a ballot: one memory round trip instead of up to
// 2 * l_steps dependent ones.
template <bool WAVE_PAR>
__device__ __forceinline__ float3 get_mid(const TplLatLonParams& p, const TplDpEnvParams& ep, const DistSrc& src,
                                          const TplLatLonState& tp, float dt)
{
    const float l_step_size = (p.l_max - p.l_min) / (float)(p.l_steps - 1);
    const float mean_dist = fmaxf((float)((double)p.length_veh * 0.5), tp.ds * dt);
    float l_left = 0.0f, l_right = 0.0f;
    if constexpr (WAVE_PAR) {
        const int ln = (int)(threadIdx.x & 63u);
        const bool right = ln >= 32;
        bool f_left = false, f_right = false;
        for (int i0 = 0; i0 < p.l_steps && !(f_left && f_right); i0 += 32) {
            const int i = i0 + (ln & 31);
            const float l = right ? tp.l - (float)i * l_step_size : tp.l + (float)i * l_step_size;
            const bool hit = i < p.l_steps && dist_at(src, ep, l) < mean_dist;
            const unsigned long long m = __ballot(hit);
            const unsigned ml = (unsigned)m, mr = (unsigned)(m >> 32);
            if (!f_left && ml) { l_left = tp.l + (float)(i0 + __ffs(ml) - 1) * l_step_size; f_left = true; }
            if (!f_right && mr) { l_right = tp.l - (float)(i0 + __ffs(mr) - 1) * l_step_size; f_right = true; }
        }
    } else {
        for (int i = 0; i < p.l_steps; ++i) {
            const float l = tp.l + (float)i * l_step_size;
            if (dist_at(src, ep, l) < mean_dist) { l_left = l; break; }
        }
        for (int i = 0; i < p.l_steps; ++i) {
            const float l = tp.l - (float)i * l_step_size;
            if (dist_at(src, ep, l) < mean_dist) { l_right = l; break; }
        }
    }
    float3 res;
    const float mid = (float)((double)l_right + (double)(l_left - l_right) * 0.5);
    res.y = fminf(l_right + p.d_lat_comf, mid);
    res.z = fmaxf(l_left - p.d_lat_comf, mid);
    if (p.l_trg < l_right) res.x = res.y;
    else if (p.l_trg > l_left) res.x = res.z;
    else res.x = p.l_trg;
    return res;
}

// Evaluator::evalState, :126-166
template <bool WAVE_PAR = false>
__device__ __forceinline__ Eval eval_state(const TplLatLonParams& p, const TplDpEnvParams& ep, const DistSrc& src,
                                           const TplLatLonState& tp, float dt, float4 rp)
{
    Eval r = {0.0f, 0.0f, 0u};
    const float v_max_ref = rp.x;
    const float d_left_ref = (float)((double)rp.y - (double)p.width_veh * 0.5);
    const float d_right_ref = (float)(-((double)rp.z - (double)p.width_veh * 0.5));
    r.cost = (float)((double)r.cost + 1000.0 * (double)fmaxf(0.0f, tp.l - d_left_ref));
    r.cost = (float)((double)r.cost + 1000.0 * (double)fmaxf(0.0f, d_right_ref - tp.l));
    const float3 mid = get_mid<WAVE_PAR>(p, ep, src, tp, dt);
    r.cost += p.w_l * sqf(mid.x - tp.l);
    if (tp.l < mid.y) r.cost += p.w_lat_dist * sqf(mid.y - tp.l);
    if (tp.l > mid.z) r.cost += p.w_lat_dist * sqf(mid.z - tp.l);
    r.cost = (float)((double)r.cost + (double)p.w_progress * (1000.0 - (double)tp.s));
    if (tp.ds > v_max_ref) { r.constr += tp.ds - v_max_ref; r.flags |= CONSTR_VELOCITY; }
    if (tp.t < p.t_st_min) r.cost += p.w_spatio_temporal * fmaxf(0.0f, tp.s - p.s_st);
    if (tp.t > p.t_st_max) r.cost += p.w_spatio_temporal * fmaxf(0.0f, p.s_st - tp.s);
    return r;
}

// Evaluator::finalState, :68-81
template <bool WAVE_PAR = false>
__device__ __forceinline__ Eval final_state(const TplLatLonParams& p, const TplDpEnvParams& ep, const DistSrc& src,
                                            const TplLatLonState& tp, float dt, float4 rp)
{
    Eval r = {0.0f, 0.0f, 0u};
    if (roundf(rp.w) == 1.0f) r.cost += p.w_xing_slow;
    const float3 mid = get_mid<WAVE_PAR>(p, ep, src, tp, dt);
    r.cost += p.w_l * sqf(mid.x - tp.l);
    return r;
}

// Evaluator::evalAction, :168-204
__device__ __forceinline__ Eval eval_action(const TplLatLonParams& p, const TplLatLonState& tp, float dt,
                                            float d_fwd, float d_safety, float dds, float dl,
                                            float tn_s, float tn_l, float tn_dds, float tn_dl)
{
    Eval r = {0.0f, 0.0f, 0u};
    const float s_change = tn_s - tp.s;
    const float l_change = tn_l - tp.l;
    const float slope = fabsf(l_change / s_change);
    if (slope > p.slope_abs_max) {
        r.constr = (float)((double)r.constr + (double)fabsf(slope - p.slope_abs_max) * 1000.0);
        r.flags |= CONSTR_ANGLE;
    }
    if (s_change > d_fwd) { r.constr += s_change - d_fwd; r.flags |= CONSTR_OCCUPANCY; }
    r.cost += p.w_safety_dist * fmaxf(0.0f, s_change - d_safety);
    const float ddds = tn_dds - dds;
    const float ddl = tn_dl - dl;
    r.cost += p.w_dds * sqf(dds * dt);
    r.cost += p.w_ddds * sqf(ddds);
    r.cost += p.w_dl * sqf(dl * dt);
    r.cost += p.w_ddl * sqf(ddl);
    return r;
}

// lateral rate of action column dl_idx, :223-246
__device__ __forceinline__ float action_dl(const TplLatLonParams& p, int dl_idx, int n_dl)
{
    const float step_size_dl = (p.dl_max - p.dl_min) / (float)(n_dl - 1);
    const int half = n_dl / 2;
    if (dl_idx == 0) return 0.0f;
    if (dl_idx <= half) return step_size_dl * (float)dl_idx;
    return -step_size_dl * (float)(dl_idx - half);
}

__device__ __forceinline__ float action_dds(const TplLatLonParams& p, int dds_idx, int n_dds)
{
    return p.dds_min + (p.dds_max - p.dds_min) * ((float)dds_idx / (float)(n_dds - 1));
}

// get_cached_scaled with a point-filtered texture (arr_tex_surf.cuh:151-158): texel floor(x + 0.5)
__device__ __forceinline__ int point_idx(float x, float x_min, float step, int n)
{
    const float c = (x - x_min) / step + 0.5f;
    return clampi((int)fminf(fmaxf(floorf(c), -1.0f), (float)n), 0, n - 1);
}

// lerp_cached_scaled (arr_tex_surf.cuh:160-167): filtered fetch under the planner's filter rule (default R2/R3: l, then ds, then s)
__device__ __forceinline__ float4 lerp_node(const float4* __restrict__ tab, const TplLatLonParams& p,
                                            float s, float ds, float l, float step_s, float step_ds, float step_l,
                                            FilterRule rule = kDefaultRule)
{
    const Tap ts = make_tap((s - p.s_min) / step_s + 0.5f, p.s_steps, rule);
    const Tap td = make_tap((ds - p.ds_min) / step_ds + 0.5f, p.ds_steps, rule);
    const Tap tl = make_tap((l - p.l_min) / step_l + 0.5f, p.l_steps, rule);
    auto at = [&](int il, int id, int is) { return tab[((size_t)il * p.ds_steps + id) * p.s_steps + is]; };
    // corners x + 2 y + 4 z of the texture: x = s, y = ds, z = l
    const float4 n[8] = {at(tl.i0, td.i0, ts.i0), at(tl.i0, td.i0, ts.i1), at(tl.i0, td.i1, ts.i0), at(tl.i0, td.i1, ts.i1),
                         at(tl.i1, td.i0, ts.i0), at(tl.i1, td.i0, ts.i1), at(tl.i1, td.i1, ts.i0), at(tl.i1, td.i1, ts.i1)};
    float cx[8], cy[8], cz[8], cw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { cx[i] = n[i].x; cy[i] = n[i].y; cz[i] = n[i].z; cw[i] = n[i].w; }
    return make_float4(blend8(cx, ts.w, td.w, tl.w, rule), blend8(cy, ts.w, td.w, tl.w, rule),
                       blend8(cz, ts.w, td.w, tl.w, rule), blend8(cw, ts.w, td.w, tl.w, rule));
}

// the 2-D filter along s (y) and t (x) on ONE lateral cell of the distance map: what the 3-D filter of dist_lon_x reduces to at
// an integral lateral coordinate (weight 0 along l) under every rule -- corners t[x + 2 y]
__device__ __forceinline__ float blend4_xy(const float (&t)[4], float wx, float wy, FilterRule rule)
{
    if (rule.order == 1) return lerp1(lerp1(t[0], t[1], wx, rule), lerp1(t[2], t[3], wx, rule), wy, rule);      // x, then y (, then z)
    if (rule.order >= 2) {
        const float x2[2] = {1.0f - wx, wx}, y2[2] = {1.0f - wy, wy};
        float w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float xy = x2[c & 1] * y2[c >> 1]; w[c] = xy * 1.0f; }
        return wsum<4>(w, t, rule);
    }
    return lerp1(lerp1(t[0], t[2], wy, rule), lerp1(t[1], t[3], wy, rule), wx, rule);                               // (z, then) y, then x
}

__global__ void k_ll_prelude(float4* __restrict__ ref_s, float* __restrict__ profile,
                             const float4* __restrict__ ref, int n_ref, float ref_step,
                             const float* __restrict__ dist_lon, TplLatLonParams p, TplDpEnvParams ep,
                             float step_s, FilterRule rule)
{
    const int idx_t = blockIdx.x;
    for (int idx_s = threadIdx.x; idx_s < p.s_steps; idx_s += blockDim.x) {
        const float s = p.s_min + (float)idx_s * step_s;
        if (idx_t == 0) ref_s[idx_s] = ref_tex(ref, n_ref, ref_step, s, rule);
        if (idx_t >= 1) {
            const float t = p.dt_start + (float)(idx_t - 1) * p.dt;
            float* out = profile + ((size_t)idx_t * p.s_steps + idx_s) * ep.l_steps;
            for (int il = 0; il < ep.l_steps; ++il) {
                // interpDistMapLon rounds the lateral coordinate, so the lookup sits ON lateral cell il:
                // its filter weight along l is exactly 0 and lerp(x, y, 0) == x for the finite
                // distances of the map
                const float t_idx = env_t_idx(ep, t);
                const float sc = (s - ep.s_min) / (ep.s_max - ep.s_min) * (float)(ep.s_steps - 1);
                const Tap tt = make_tap(t_idx + 0.5f, ep.t_steps, rule);
                const Tap ts = make_tap(sc + 0.5f, ep.s_steps, rule);
                auto at = [&](int is, int it) { return dist_lon[(((size_t)il * ep.s_steps + is) * ep.t_steps + it) * 2]; };
                const float t4[4] = {at(ts.i0, tt.i0), at(ts.i0, tt.i1), at(ts.i1, tt.i0), at(ts.i1, tt.i1)};     // x = t, y = s
                out[il] = blend4_xy(t4, tt.w, ts.w, rule);
            }
        }
    }
}

// ------------------------------------------------------------------------
// backwardStep, lat_lon_planner.cu:276-327.  grid (ds_steps, l_steps), lanes over s.
// tables: float4[l][ds][s].
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_ll_backward(float4* __restrict__ nodes, const float4* __restrict__ next, int idx_t,
                              const float4* __restrict__ ref_s, const float* __restrict__ profile,
                              TplLatLonParams p, TplDpEnvParams ep, float step_s, float step_ds, float step_l)
{
    // the (s, ds) cells of one lateral offset, flattened: every wavefront is full (rows of 201
    // stations would leave a fifth of the lanes of a 256-thread row block idle)
    const int idx_l = blockIdx.y;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell < p.s_steps * p.ds_steps) {
        const int idx_ds = cell / p.s_steps, idx_s = cell - idx_ds * p.s_steps;
        TplLatLonState tp;
        tp.t = p.dt_start + (float)(idx_t - 1) * p.dt;
        tp.s = p.s_min + (float)idx_s * step_s;
        tp.ds = p.ds_min + (float)idx_ds * step_ds;
        tp.l = p.l_min + (float)idx_l * step_l;
        tp.dds = 0.0f; tp.ddds = 0.0f; tp.dl = 0.0f; tp.ddl = 0.0f; tp.dddl = 0.0f;
        tp.cost = 0.0f; tp.constr = 0.0f; tp.flags_constr = 0;
        const float dt = p.dt;
        const float4 rp = ref_s[idx_s];
        DistSrc src;
        src.profile = profile + ((size_t)idx_t * p.s_steps + idx_s) * ep.l_steps;
        src.dist_lon = nullptr; src.t = tp.t; src.s = tp.s;

        const Eval st = eval_state(p, ep, src, tp, dt, rp);
        float tot_cost, tot_constr, best_dds = 0.0f, best_dl = 0.0f;
        if (idx_t < p.t_steps - 1) {
            // findBestAction<7, 7>, :206-273.  The successor's station and speed depend on the
            // longitudinal action only, its offset on the lateral one: 7 + 7 + 7 index computations
            // (a division each) instead of 3 x 49, and the seven table cells of a lateral action
            // are requested together, one lateral action ahead of their use.
            constexpr int NDDS = 7, NDL = 7;
            tot_cost = INFINITY; tot_constr = INFINITY;
            float a_dds[NDDS], tn_s[NDDS];
            int base[NDDS];
#pragma unroll
            for (int k = 0; k < NDDS; ++k) {
                a_dds[k] = action_dds(p, k, NDDS);
                const TplLatLonState tn = ll_dynamics(tp, a_dds[k], 0.0f, dt);
                tn_s[k] = tn.s;
                base[k] = point_idx(tn.ds, p.ds_min, step_ds, p.ds_steps) * p.s_steps
                          + point_idx(tn.s, p.s_min, step_s, p.s_steps);
            }
            float d_fwd_left = 0.0f, d_fwd_right = 0.0f;
            float4 nn[NDDS];
            {
                const int row0 = point_idx(tp.l, p.l_min, step_l, p.l_steps) * p.ds_steps * p.s_steps;
#pragma unroll
                for (int k = 0; k < NDDS; ++k) nn[k] = next[(size_t)(row0 + base[k])];
            }
            for (int dl_idx = 0; dl_idx < NDL; ++dl_idx) {
                const float dl = action_dl(p, dl_idx, NDL);
                const float tn_l = tp.l + dl * dt;
                float d_fwd;
                if (dl_idx == 0) {
                    const float d = dist_at(src, ep, tp.l);
                    d_fwd_left = d; d_fwd_right = d; d_fwd = d;
                } else if (dl_idx <= NDL / 2) {
                    const float d = dist_at(src, ep, tn_l);
                    d_fwd_left = fminf(d, d_fwd_left);
                    d_fwd = d_fwd_left;
                } else {
                    const float d = dist_at(src, ep, tn_l);
                    d_fwd_right = fminf(d, d_fwd_right);
                    d_fwd = d_fwd_right;
                }
                d_fwd = (float)((double)d_fwd - (double)p.length_veh * 0.5);
                const float d_safety = d_fwd - p.gap_min - tp.ds * p.time_gap;
                // the cells of the NEXT lateral action fly while this one is evaluated
                float4 cur[NDDS];
#pragma unroll
                for (int k = 0; k < NDDS; ++k) cur[k] = nn[k];
                if (dl_idx + 1 < NDL) {
                    const float l1 = tp.l + action_dl(p, dl_idx + 1, NDL) * dt;
                    const int row1 = point_idx(l1, p.l_min, step_l, p.l_steps) * p.ds_steps * p.s_steps;
#pragma unroll
                    for (int k = 0; k < NDDS; ++k) nn[k] = next[(size_t)(row1 + base[k])];
                }
#pragma unroll
                for (int dds_idx = 0; dds_idx < NDDS; ++dds_idx) {
                    const float dds = a_dds[dds_idx];
                    const float4 v = cur[dds_idx];
                    const Eval a = eval_action(p, tp, dt, d_fwd, d_safety, dds, dl, tn_s[dds_idx], tn_l, v.z, v.w);
                    const float c = a.cost + v.x, g = a.constr + v.y;
                    if (eval_less(c, g, tot_cost, tot_constr)) {
                        tot_cost = c; tot_constr = g;
                        best_dds = dds; best_dl = dl;
                    }
                }
            }
        } else {
            const Eval f = final_state(p, ep, src, tp, dt, rp);
            tot_cost = f.cost; tot_constr = f.constr;
        }
        nodes[((size_t)idx_l * p.ds_steps + idx_ds) * p.s_steps + idx_s] =
            make_float4(st.cost + tot_cost, st.constr + tot_constr, best_dds, best_dl);
    }
}

// ------------------------------------------------------------------------
// forwardStep x t_steps, lat_lon_planner.cu:329-355, 613-632, in ONE launch: a workgroup of seven
// wavefronts, one of the 21 x 21 actions of a stage per thread.  A stage is a handful of memory
// round trips -- reference line, the lateral clearance walk (get_mid<true>), the 21 forward
// distances, the eight table cells of every action, all requested side by side -- and the state
// moves from stage to stage in registers (every thread applies the chosen action), not through
// memory.  traj: LatLonState[t_steps], traj[0] = start state.
// ------------------------------------------------------------------------
constexpr int LL_FWD_THREADS = 448;                     // 441 actions, 7 wavefronts

__global__ __launch_bounds__(LL_FWD_THREADS)
void k_ll_forward(TplLatLonState* __restrict__ traj, const float4* __restrict__ tables, size_t table_cells,
                  const float4* __restrict__ ref, int n_ref, float ref_step,
                  const float* __restrict__ dist_lon, TplLatLonParams p, TplDpEnvParams ep,
                  float step_s, float step_ds, float step_l, FilterRule rule)
{
    constexpr int NDDS = 21, NDL = 21, NW = LL_FWD_THREADS / 64;
    __shared__ float sh_dfwd[NDL];
    __shared__ float sh_c[NW], sh_g[NW], sh_ac[NW], sh_ag[NW];
    __shared__ unsigned sh_fl[NW];
    __shared__ int sh_a[NW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    TplLatLonState tp = traj[0];
    for (int idx_t = 0; idx_t < p.t_steps; ++idx_t) {
        const float dt = (idx_t == 0) ? p.dt_start : p.dt;
        const float4 rp = ref_tex(ref, n_ref, ref_step, tp.s, rule);
        DistSrc src;
        src.profile = nullptr; src.dist_lon = dist_lon; src.t = tp.t; src.s = tp.s; src.rule = rule;
        const Eval st = eval_state<true>(p, ep, src, tp, dt, rp);  // every thread, same result
        tp.cost = st.cost; tp.constr = st.constr; tp.flags_constr = (uint8_t)st.flags;
        if (idx_t < p.t_steps - 1) {
            const float4* next = tables + (size_t)min(p.t_steps - 1, idx_t + 1) * table_cells;
            // forward distances of the 21 lateral rates; running minima per side as in :223-246
            if (tid < NDL) sh_dfwd[tid] = dist_at(src, ep, tp.l + action_dl(p, tid, NDL) * dt);
            __syncthreads();
            float best_c = INFINITY, best_g = INFINITY, best_ac = 0.0f, best_ag = 0.0f;
            unsigned best_fl = 0u;
            int best_a = 0x7fffffff;
            if (tid < NDL * NDDS) {
                const int a = tid;
                const int dl_idx = a / NDDS, dds_idx = a - dl_idx * NDDS;
                const float dl = action_dl(p, dl_idx, NDL);
                float d_fwd = sh_dfwd[0];
                if (dl_idx >= 1 && dl_idx <= NDL / 2) {
                    for (int k = 1; k <= dl_idx; ++k) d_fwd = fminf(sh_dfwd[k], d_fwd);
                } else if (dl_idx > NDL / 2) {
                    for (int k = NDL / 2 + 1; k <= dl_idx; ++k) d_fwd = fminf(sh_dfwd[k], d_fwd);
                }
                d_fwd = (float)((double)d_fwd - (double)p.length_veh * 0.5);
                const float d_safety = d_fwd - p.gap_min - tp.ds * p.time_gap;
                const float dds = action_dds(p, dds_idx, NDDS);
                const TplLatLonState tn = ll_dynamics(tp, dds, dl, dt);
                const float4 nn = lerp_node(next, p, tn.s, tn.ds, tn.l, step_s, step_ds, step_l, rule);
                const Eval ev = eval_action(p, tp, dt, d_fwd, d_safety, dds, dl, tn.s, tn.l, nn.z, nn.w);
                const float c = ev.cost + nn.x, g = ev.constr + nn.y;
                if (eval_less(c, g, best_c, best_g)) {
                    best_c = c; best_g = g; best_ac = ev.cost; best_ag = ev.constr; best_fl = ev.flags; best_a = a;
                }
            }
            // argmin: smaller (constr, cost), ties -> the action that comes first in loop order;
            // inside each wavefront by butterflies, across the seven through LDS
            for (int off = 32; off > 0; off >>= 1) {
                const float oc = __shfl_xor(best_c, off, 64), og = __shfl_xor(best_g, off, 64);
                const float oac = __shfl_xor(best_ac, off, 64), oag = __shfl_xor(best_ag, off, 64);
                const unsigned ofl = __shfl_xor(best_fl, off, 64);
                const int oa = __shfl_xor(best_a, off, 64);
                const bool take = eval_less(oc, og, best_c, best_g)
                                  || (!eval_less(best_c, best_g, oc, og) && oa < best_a);
                if (take) { best_c = oc; best_g = og; best_ac = oac; best_ag = oag; best_fl = ofl; best_a = oa; }
            }
            if (lane == 0) {
                sh_c[wave] = best_c; sh_g[wave] = best_g; sh_ac[wave] = best_ac; sh_ag[wave] = best_ag;
                sh_fl[wave] = best_fl; sh_a[wave] = best_a;
            }
            __syncthreads();
            best_c = sh_c[0]; best_g = sh_g[0]; best_ac = sh_ac[0]; best_ag = sh_ag[0]; best_fl = sh_fl[0]; best_a = sh_a[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float oc = sh_c[w], og = sh_g[w];
                const int oa = sh_a[w];
                const bool take = eval_less(oc, og, best_c, best_g)
                                  || (!eval_less(best_c, best_g, oc, og) && oa < best_a);
                if (take) { best_c = oc; best_g = og; best_ac = sh_ac[w]; best_ag = sh_ag[w]; best_fl = sh_fl[w]; best_a = oa; }
            }
            if (best_a != 0x7fffffff) {
                const int dl_idx = best_a / NDDS, dds_idx = best_a - dl_idx * NDDS;
                tp.dds = action_dds(p, dds_idx, NDDS);
                tp.dl = action_dl(p, dl_idx, NDL);
                tp.cost += best_ac;
                tp.constr += best_ag;
                tp.flags_constr |= (uint8_t)best_fl;
            } else {
                // no action compared below (inf, inf): minRes keeps its default and dds/dl their old values
                tp.cost += 0.0f; tp.constr += 0.0f;
            }
            const TplLatLonState tn = ll_dynamics(tp, tp.dds, tp.dl, dt);
            if (tid == 0) {
                traj[idx_t] = tp;
                traj[idx_t + 1] = tn;
            }
            tp = tn;                                     // what the next stage would read back
            __syncthreads();                             // sh_* are rewritten by the next stage
        } else if (tid == 0) {
            traj[idx_t] = tp;
        }
    }
}

// reevalTraj, lat_lon_planner.cu:357-404: one thread per node (a node only writes itself)
__global__ void k_ll_reeval(TplLatLonState* __restrict__ out, const TplLatLonState* __restrict__ in, int len,
                            const float4* __restrict__ ref, int n_ref, float ref_step,
                            const float* __restrict__ dist_lon, TplLatLonParams p, TplDpEnvParams ep, FilterRule rule)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= len) return;
    TplLatLonState tp = in[idx];
    const float dt = (idx == len - 1) ? 0.0f : in[idx + 1].t - tp.t;
    const float4 rp = ref_tex(ref, n_ref, ref_step, tp.s, rule);
    DistSrc src;
    src.profile = nullptr; src.dist_lon = dist_lon; src.t = tp.t; src.s = tp.s; src.rule = rule;
    const Eval st = eval_state(p, ep, src, tp, dt, rp);
    tp.cost = st.cost; tp.constr = st.constr; tp.flags_constr = (uint8_t)st.flags;
    const float l_next = tp.l + dt * tp.dl;
    const float l_step_size = (p.l_max - p.l_min) / (float)(p.l_steps - 1);
    const float l_dist = l_next - tp.l;
    const int steps = (int)ceilf(fabsf(l_dist) / l_step_size);
    const float step_size = l_dist / (float)steps;
    float d_fwd = dist_at(src, ep, tp.l);
    for (int i = 0; i < steps; ++i) {
        const float l = tp.l + (float)i * step_size;
        d_fwd = fminf(dist_at(src, ep, l), d_fwd);
    }
    const float d_safety = d_fwd - p.gap_min - tp.ds * p.time_gap;
    if (idx < len - 1) {
        const TplLatLonState tn = in[idx + 1];
        const Eval a = eval_action(p, tp, dt, d_fwd, d_safety, tp.dds, tp.dl, tn.s, tn.l, tn.dds, tn.dl);
        tp.cost += a.cost; tp.constr += a.constr; tp.flags_constr |= (uint8_t)a.flags;
    }
    out[idx] = tp;
}

__global__ void k_ll_query(float4* __restrict__ out, const float4* __restrict__ tab, float s, float ds, float l,
                           TplLatLonParams p, float step_s, float step_ds, float step_l)
{
    const int is = point_idx(s, p.s_min, step_s, p.s_steps);
    const int id = point_idx(ds, p.ds_min, step_ds, p.ds_steps);
    const int il = point_idx(l, p.l_min, step_l, p.l_steps);
    out[0] = tab[((size_t)il * p.ds_steps + id) * p.s_steps + is];
}

// ---- host: InterpVars on the float member t, utils.hpp:59-86 ----------------------------------

template <typename S>
void bracket(const S* v, int n, double t, int* i_prev, int* i_next, double* a)
{
    *i_prev = 0; *i_next = 0;
    if (t < (double)v[0].t) { *i_prev = 0; *i_next = 1; }
    else if (t >= (double)v[n - 1].t) { *i_prev = n - 2; *i_next = n - 1; }
    else {
        for (int i = 1; i < n; ++i)
            if ((double)v[i - 1].t <= t && t < (double)v[i].t) { *i_prev = i - 1; *i_next = i; break; }
    }
    const double tp = (double)v[*i_prev].t, tn = (double)v[*i_next].t;
    *a = std::max(0.0, std::min(1.0, (t - tp) / (tn - tp)));
}

// lqrSmoother<4, 1> with constant lxx, luu, utils.hpp:278-348 (plain loops instead of Eigen)
void lqr_smooth4(const double x0[4], const std::vector<std::array<double, 4>>& x_ref, const double fx[4][4],
                 const double fu[4], const double lxx[4][4], double luu, std::vector<std::array<double, 4>>& xs)
{
    const int H = (int)x_ref.size();
    std::vector<double> k((size_t)H, 0.0);
    std::vector<std::array<double, 4>> K((size_t)H, std::array<double, 4>{0, 0, 0, 0});
    double Vxx[4][4], Vx[4];
    for (int i = 0; i < 4; ++i) {
        double acc = 0.0;
        for (int j = 0; j < 4; ++j) { Vxx[i][j] = lxx[i][j]; acc += lxx[i][j] * x_ref[(size_t)H - 1][j]; }
        Vx[i] = -acc;
    }
    for (int it = H - 2; it >= 0; --it) {
        double lx[4];
        for (int i = 0; i < 4; ++i) {
            double acc = 0.0;
            for (int j = 0; j < 4; ++j) acc += lxx[i][j] * x_ref[(size_t)it][j];
            lx[i] = -acc;
        }
        // products associate left to right as written in the reference: (fx^T * Vxx) * fx etc.
        double Qx[4], Qu = 0.0, Qxx[4][4], Quu, Qux[4];
        double FtV[4][4], fuV[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int m = 0; m < 4; ++m) acc += fx[m][i] * Vxx[m][j];
                FtV[i][j] = acc;
            }
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += fu[m] * Vxx[m][j];
            fuV[j] = acc;
        }
        for (int i = 0; i < 4; ++i) {
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += fx[m][i] * Vx[m];
            Qx[i] = lx[i] + acc;
            Qu += fu[i] * Vx[i];
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int m = 0; m < 4; ++m) acc += FtV[i][m] * fx[m][j];
                Qxx[i][j] = lxx[i][j] + acc;
            }
        {
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += fuV[m] * fu[m];
            Quu = luu + acc;
        }
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int m = 0; m < 4; ++m) acc += fuV[m] * fx[m][j];
            Qux[j] = acc;
        }
        const double W = -1.0 / Quu;
        k[(size_t)it] = W * Qu;
        for (int j = 0; j < 4; ++j) K[(size_t)it][(size_t)j] = W * Qux[j];
        const double kk = k[(size_t)it];
        const std::array<double, 4>& Kk = K[(size_t)it];
        for (int i = 0; i < 4; ++i) Vx[i] = Qx[i] + Kk[(size_t)i] * Quu * kk + Kk[(size_t)i] * Qu + Qux[i] * kk;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const double tmp_ij = Kk[(size_t)i] * Qux[j], tmp_ji = Kk[(size_t)j] * Qux[i];
                Vxx[i][j] = Qxx[i][j] + Kk[(size_t)i] * Quu * Kk[(size_t)j] + (tmp_ij + tmp_ji);
            }
    }
    xs.assign((size_t)H, std::array<double, 4>{0, 0, 0, 0});
    for (int i = 0; i < 4; ++i) xs[0][(size_t)i] = x0[i];
    for (int it = 0; it < H - 1; ++it) {
        double u = k[(size_t)it];
        for (int j = 0; j < 4; ++j) u += K[(size_t)it][(size_t)j] * xs[(size_t)it][(size_t)j];
        for (int i = 0; i < 4; ++i) {
            double acc = 0.0;
            for (int j = 0; j < 4; ++j) acc += fx[i][j] * xs[(size_t)it][(size_t)j];
            xs[(size_t)it + 1][(size_t)i] = acc + fu[i] * u;
        }
    }
}

} // namespace

struct tplhip_lat_lon {
    int device = 0;
    hipStream_t stream = nullptr;
    TplLatLonParams params;
    bool have_params = false;
    float4* d_tables = nullptr;         // [t][l][ds][s]
    size_t table_cells = 0;
    float4* d_ref_s = nullptr;          // [s]
    float* d_profile = nullptr;         // [t][s][env l]
    size_t profile_cap = 0;
    float4* d_ref = nullptr;            // reference line texture, [n_ref]
    size_t ref_cap = 0;
    TplLatLonState* d_traj = nullptr;   // [max(t_steps, reeval length)]
    TplLatLonState* d_traj2 = nullptr;
    size_t traj_cap = 0;
    float4* d_q = nullptr;
    bool tables_valid = false;
    FilterRule rule = kDefaultRule;     // the texture filter's rules (tplhip_lat_lon_set_filter_rule)
};

namespace {

void ll_steps(const TplLatLonParams& p, float* ss, float* sd, float* sl)     // ArrTexSurf::rescale, arr_tex_surf.cuh:121-135
{
    *ss = (p.s_max - p.s_min) / (float)(p.s_steps - 1);
    *sd = (p.ds_max - p.ds_min) / (float)(p.ds_steps - 1);
    *sl = (p.l_max - p.l_min) / (float)(p.l_steps - 1);
}

int ll_check_env(tplhip_lat_lon* h, tplhip_dp_env* env)
{
    if (!h || !env) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (!h->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the planner");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    if (!env->rendered || env->ref_line.empty())
        return tplhip_fail(TPLHIP_ESTATE, "the environment was never updated (no distance map / reference line)");
    if (env->device != h->device) return tplhip_fail(TPLHIP_EINVAL, "planner and environment live on different devices");
    return TPLHIP_OK;
}

int ll_upload_ref(tplhip_lat_lon* h, tplhip_dp_env* env)
{
    // RefLineGpu::upload, common.cuh:59-78: (v_max, d_left, d_right, semantic) as floats
    const size_t n = env->ref_line.size();
    if (n > h->ref_cap) {
        (void)hipFree(h->d_ref); h->d_ref = nullptr; h->ref_cap = 0;
        HIP_TRY(hipMalloc(&h->d_ref, sizeof(float4) * n), "ref line texture allocation failed");
        h->ref_cap = n;
    }
    std::vector<float4> tex(n);
    for (size_t i = 0; i < n; ++i)
        tex[i] = make_float4((float)env->ref_line[i].v_max, (float)env->ref_line[i].d_left,
                             (float)env->ref_line[i].d_right, (float)env->ref_line[i].semantic);
    HIP_TRY(hipMemcpyAsync(h->d_ref, tex.data(), sizeof(float4) * n, hipMemcpyHostToDevice, h->stream),
            "copying ref line to texture failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "copying ref line to texture failed");
    return TPLHIP_OK;
}

int ll_ensure_traj(tplhip_lat_lon* h, size_t n)
{
    if (n <= h->traj_cap && h->d_traj) return TPLHIP_OK;
    (void)hipFree(h->d_traj); (void)hipFree(h->d_traj2);
    h->d_traj = nullptr; h->d_traj2 = nullptr; h->traj_cap = 0;
    HIP_TRY(hipMalloc(&h->d_traj, sizeof(TplLatLonState) * n), "forward nodes allocation failed");
    HIP_TRY(hipMalloc(&h->d_traj2, sizeof(TplLatLonState) * n), "reeval nodes allocation failed");
    h->traj_cap = n;
    return TPLHIP_OK;
}

} // namespace

extern "C" {

void tplhip_lat_lon_params_default(TplLatLonParams* p)
{
    // lat_lon_planner.cuh:11-61
    p->s_min = 0.0f; p->s_max = 200.0f; p->ds_min = 0.0f; p->ds_max = 36.0f; p->l_min = -5.0f; p->l_max = 5.0f;
    p->dds_min = -2.0f; p->dds_max = 2.0f; p->dl_min = -2.0f; p->dl_max = 2.0f;
    p->t_steps = 10; p->s_steps = 201; p->ds_steps = 37; p->l_steps = 21;
    p->dt = 1.0f; p->dt_start = 1.0f; p->dt_smooth_traj = 0.1f;
    p->dds_start = 0.0f; p->w_dds_start = 10.0f; p->angle_start = 0.0f; p->w_angle_start = 10.0f;
    p->l_trg = 0.0f;
    p->w_progress = 1.0f; p->w_dds = 1.0f; p->w_ddds = 1.0f; p->w_l = 1.0f; p->w_dl = 1.0f; p->w_ddl = 1.0f;
    p->w_safety_dist = 10.0f; p->w_xing_slow = 1.0f;
    p->slope_abs_max = 0.8f;
    p->w_lat_dist = 0.0f; p->d_lat_comf = 2.0f;
    p->time_gap = 2.0f; p->gap_min = 2.0f;
    p->t_st_min = -1.0f; p->t_st_max = -1.0f; p->s_st = 0.0f; p->w_spatio_temporal = 10.0f;
    p->width_veh = 2.0f; p->length_veh = 6.0f;
}

void tplhip_lat_lon_dynamics(const TplLatLonState* x0, float dds, float dl, float dt, TplLatLonState* out)
{
    *out = ll_dynamics(*x0, dds, dl, dt);
}

int tplhip_lat_lon_traj_state(const TplLatLonState* st, int n, float t, TplLatLonState* out)
{
    // LatLonTraj::state, lat_lon_planner.cu:432-441
    if (!st || !out || n < 2) return tplhip_fail(TPLHIP_EINVAL, "trajectory needs at least two states");
    int ip, in_; double a;
    bracket(st, n, (double)t, &ip, &in_, &a);
    const float t_rel = t - st[ip].t;
    *out = ll_dynamics(st[ip], st[ip].dds, st[ip].dl, t_rel);
    return TPLHIP_OK;
}

int tplhip_lat_lon_traj_lerp(const TplLatLonState* st, int n, float t, TplLatLonState* out)
{
    // LatLonTraj::lerp, :443-463
    if (!st || !out || n < 2) return tplhip_fail(TPLHIP_EINVAL, "trajectory needs at least two states");
    int ip, in_; double a;
    bracket(st, n, (double)t, &ip, &in_, &a);
    const float ai = (float)(1.0 - a);
    const TplLatLonState& p = st[ip];
    const TplLatLonState& q = st[in_];
    TplLatLonState r;
    std::memset(&r, 0, sizeof r);
#define LL_MIX(f) r.f = (float)((double)(p.f * ai) + (double)q.f * a)
    LL_MIX(t); LL_MIX(s); LL_MIX(ds); LL_MIX(dds); LL_MIX(ddds); LL_MIX(l); LL_MIX(dl); LL_MIX(ddl); LL_MIX(dddl);
    LL_MIX(cost); LL_MIX(constr);
#undef LL_MIX
    *out = r;
    return TPLHIP_OK;
}

int tplhip_lat_lon_cart_lerp(const TplLatLonCartState* st, int n, float t, TplLatLonCartState* out)
{
    // LatLonCartTraj::lerp, :465-486
    if (!st || !out || n < 2) return tplhip_fail(TPLHIP_EINVAL, "trajectory needs at least two states");
    int ip = 0, in_ = 0;
    const double tv = (double)t;
    if (tv < st[0].t) { ip = 0; in_ = 1; }
    else if (tv >= st[n - 1].t) { ip = n - 2; in_ = n - 1; }
    else {
        for (int i = 1; i < n; ++i)
            if (st[i - 1].t <= tv && tv < st[i].t) { ip = i - 1; in_ = i; break; }
    }
    double a = (tv - st[ip].t) / (st[in_].t - st[ip].t);
    a = std::max(0.0, std::min(1.0, a));
    const float ai = (float)(1.0 - a);
    const TplLatLonCartState& p = st[ip];
    const TplLatLonCartState& q = st[in_];
    TplLatLonCartState r;
    std::memset(&r, 0, sizeof r);
#define CART_MIX(f) r.f = p.f * (double)ai + q.f * a
    CART_MIX(t); CART_MIX(distance); CART_MIX(x); CART_MIX(y); CART_MIX(v); CART_MIX(a);
    r.heading = p.heading + short_angle_dist(p.heading, q.heading) * a;
    CART_MIX(k); CART_MIX(cost); CART_MIX(constr);
#undef CART_MIX
    *out = r;
    return TPLHIP_OK;
}

int tplhip_lat_lon_create(int device, tplhip_lat_lon** out)
{
    if (!out) return tplhip_fail(TPLHIP_EINVAL, "null out pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return tplhip_fail(TPLHIP_EDEVICE, "no usable HIP device");
    if (device < 0) device = tplhip_default_device(n);
    if (device >= n) return tplhip_fail(TPLHIP_EDEVICE, "no usable HIP device (requested %d)", device);
    tplhip_lat_lon* h = new (std::nothrow) tplhip_lat_lon();
    if (!h) return tplhip_fail(TPLHIP_EDEVICE, "out of host memory");
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return tplhip_fail(TPLHIP_EDEVICE, "stream creation failed");
    }
    tplhip_lat_lon_params_default(&h->params);
    *out = h;
    return TPLHIP_OK;
}

int tplhip_lat_lon_destroy(tplhip_lat_lon* h)
{
    if (!h) return TPLHIP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    (void)hipFree(h->d_tables); (void)hipFree(h->d_ref_s); (void)hipFree(h->d_profile); (void)hipFree(h->d_ref);
    (void)hipFree(h->d_traj); (void)hipFree(h->d_traj2); (void)hipFree(h->d_q);
    delete h;
    return TPLHIP_OK;
}

int tplhip_lat_lon_reinit(tplhip_lat_lon* h, const TplLatLonParams* ps, int force)
{
    if (!h || !ps) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (ps->t_steps < 2 || ps->s_steps < 2 || ps->ds_steps < 2 || ps->l_steps < 2)
        return tplhip_fail(TPLHIP_EINVAL, "planner needs t_steps, s_steps, ds_steps, l_steps >= 2");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const bool full = force || !h->have_params || ps->t_steps != h->params.t_steps || ps->s_steps != h->params.s_steps
                      || ps->ds_steps != h->params.ds_steps || ps->l_steps != h->params.l_steps;
    h->params = *ps;
    h->have_params = true;
    if (!full) return TPLHIP_OK;
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    (void)hipFree(h->d_tables); (void)hipFree(h->d_ref_s); (void)hipFree(h->d_q);
    h->d_tables = nullptr; h->d_ref_s = nullptr; h->d_q = nullptr;
    h->tables_valid = false;
    h->table_cells = (size_t)ps->s_steps * ps->ds_steps * ps->l_steps;
    HIP_TRY(hipMalloc(&h->d_tables, sizeof(float4) * h->table_cells * (size_t)ps->t_steps), "backward node allocation failed");
    HIP_TRY(hipMemsetAsync(h->d_tables, 0, sizeof(float4) * h->table_cells * (size_t)ps->t_steps, h->stream), "node clear failed");
    HIP_TRY(hipMalloc(&h->d_ref_s, sizeof(float4) * (size_t)ps->s_steps), "prelude allocation failed");
    HIP_TRY(hipMalloc(&h->d_q, sizeof(float4)), "query buffer allocation failed");
    return ll_ensure_traj(h, (size_t)ps->t_steps);
}

int tplhip_lat_lon_update_traj_dp(tplhip_lat_lon* h, tplhip_dp_env* env, TplLatLonState* states)
{
    int rc = ll_check_env(h, env);
    if (rc) return rc;
    if (!states) return tplhip_fail(TPLHIP_EINVAL, "null trajectory");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const TplLatLonParams& p = h->params;
    const TplDpEnvParams& ep = env->params;
    if (p.s_steps > 1024 * 64) return tplhip_fail(TPLHIP_EINVAL, "s_steps too large");
    if ((rc = ll_upload_ref(h, env))) return rc;
    const size_t n_prof = (size_t)p.t_steps * p.s_steps * ep.l_steps;
    if (n_prof > h->profile_cap) {
        (void)hipFree(h->d_profile); h->d_profile = nullptr; h->profile_cap = 0;
        HIP_TRY(hipMalloc(&h->d_profile, sizeof(float) * n_prof), "prelude allocation failed");
        h->profile_cap = n_prof;
    }
    float ss, sd, sl;
    ll_steps(p, &ss, &sd, &sl);
    const int n_ref = (int)env->ref_line.size();
    HIP_TRY(hipMemcpyAsync(h->d_traj, states, sizeof(TplLatLonState) * (size_t)p.t_steps, hipMemcpyHostToDevice, h->stream),
            "copying trajectory to gpu failed");
    hipLaunchKernelGGL(k_ll_prelude, dim3(p.t_steps), dim3(256), 0, h->stream,
                       h->d_ref_s, h->d_profile, h->d_ref, n_ref, env->ref_step, env->d_dist_lon, p, ep, ss, h->rule);
    const int threads = 256;                                             // k_ll_backward's launch bound
    const int row_blocks = (p.s_steps * p.ds_steps + threads - 1) / threads;
    for (int i = p.t_steps - 1; i > 0; --i) {
        const int nx = std::min(p.t_steps - 1, i + 1);
        hipLaunchKernelGGL(k_ll_backward, dim3(row_blocks, p.l_steps), dim3(threads), 0, h->stream,
                           h->d_tables + (size_t)i * h->table_cells, h->d_tables + (size_t)nx * h->table_cells, i,
                           h->d_ref_s, h->d_profile, p, ep, ss, sd, sl);
    }
    hipLaunchKernelGGL(k_ll_forward, dim3(1), dim3(LL_FWD_THREADS), 0, h->stream,
                       h->d_traj, h->d_tables, h->table_cells, h->d_ref, n_ref, env->ref_step, env->d_dist_lon,
                       p, ep, ss, sd, sl, h->rule);
    HIP_TRY(hipGetLastError(), "lon planner backward pass failed");
    HIP_TRY(hipMemcpyAsync(states, h->d_traj, sizeof(TplLatLonState) * (size_t)p.t_steps, hipMemcpyDeviceToHost, h->stream),
            "copying trajectory to host failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "lon planner forward pass failed");
    h->tables_valid = true;
    return TPLHIP_OK;
}

// The texture filter's rules (R2 weight rule, R3 blend form and order: FilterRule, dp_lon_math.h) for every filtered fetch of this
// planner -- the forward pass' value-table fetch, the distance-map and reference-line look-ups of the prelude, the rollout and
// the re-evaluation.  oracle/lat_lon_oracle.c built -DLLO_WEIGHT_QUANT / -DLLO_LERP_FORM / -DLLO_BLEND_ORDER is the same choice.
int tplhip_lat_lon_set_filter_rule(tplhip_lat_lon* h, int weight_quant, int lerp_form, int blend_order)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (weight_quant < 0 || weight_quant > 2 || lerp_form < 0 || lerp_form > 2 || blend_order < 0 || blend_order > 3)
        return tplhip_fail(TPLHIP_EINVAL, "filter rule: weight_quant in {0 none, 1 rint, 2 floor}, lerp_form in {0, 1, 2}, blend_order in {0, 1, 2, 3}");
    h->rule = FilterRule{weight_quant, lerp_form, 0, blend_order};
    h->tables_valid = false;
    return TPLHIP_OK;
}

int tplhip_lat_lon_update_traj_smooth(tplhip_lat_lon* h, const TplLatLonState* traj_dp, int n_dp,
                                      const TplLatLonState* smooth0, TplLatLonState* out, int cap, int* n_out)
{
    if (!h || !traj_dp || !smooth0 || !out || !n_out || n_dp < 2) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    if (!h->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the planner");
    const TplLatLonParams& p = h->params;
    // resample, :649-655
    const int H = (int)(traj_dp[n_dp - 1].t / p.dt_smooth_traj);
    if (H < 2) return tplhip_fail(TPLHIP_EINVAL, "trajectory too short to smooth (%d samples)", H);
    if (H > cap) return tplhip_fail(TPLHIP_EINVAL, "output holds %d states, %d needed", cap, H);
    std::vector<std::array<double, 4>> ref_s((size_t)H), ref_l((size_t)H);
    for (int i = 0; i < H; ++i) {
        const float t = (float)i * p.dt_smooth_traj;
        TplLatLonState tp;
        tplhip_lat_lon_traj_state(traj_dp, n_dp, t, &tp);
        ref_s[(size_t)i] = {(double)tp.s, (double)tp.ds, 0.0, 0.0};
        ref_l[(size_t)i] = {(double)tp.l, 0.0, 0.0, 0.0};
    }
    const double x0_s[4] = {smooth0->s, smooth0->ds, smooth0->dds, smooth0->ddds};
    const double x0_l[4] = {smooth0->l, smooth0->dl, smooth0->ddl, smooth0->dddl};
    double A[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    A[0][1] = p.dt_smooth_traj; A[1][2] = p.dt_smooth_traj; A[2][3] = p.dt_smooth_traj;
    const double B[4] = {0.0, 0.0, 0.0, (double)p.dt_smooth_traj};
    double Qs[4][4] = {{0}}, Ql[4][4] = {{0}};
    Qs[0][0] = 10.0; Qs[1][1] = 10.0; Qs[2][2] = 10.0; Qs[3][3] = 10.0;
    Ql[0][0] = 1000.0; Ql[1][1] = 10.0; Ql[2][2] = 0.0; Ql[3][3] = 0.0;
    std::vector<std::array<double, 4>> xs_s, xs_l;
    lqr_smooth4(x0_s, ref_s, A, B, Qs, 1.0, xs_s);
    lqr_smooth4(x0_l, ref_l, A, B, Ql, 0.1, xs_l);
    for (int i = 0; i < H; ++i) {
        TplLatLonState tp;
        std::memset(&tp, 0, sizeof tp);
        tp.t = (float)i * p.dt_smooth_traj;
        tp.s = (float)xs_s[(size_t)i][0]; tp.ds = (float)xs_s[(size_t)i][1];
        tp.dds = (float)xs_s[(size_t)i][2]; tp.ddds = (float)xs_s[(size_t)i][3];
        tp.l = (float)xs_l[(size_t)i][0]; tp.dl = (float)xs_l[(size_t)i][1];
        tp.ddl = (float)xs_l[(size_t)i][2]; tp.dddl = (float)xs_l[(size_t)i][3];
        out[i] = tp;
    }
    *n_out = H;
    return TPLHIP_OK;
}

int tplhip_lat_lon_frenet_to_cartesian(tplhip_dp_env* env, const TplLatLonState* traj, int n, TplLatLonCartState* out)
{
    // frenetToCartesian, :768-817
    if (!env || !traj || !out || n < 0) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    if (env->ref_line.empty()) return tplhip_fail(TPLHIP_ESTATE, "the environment holds no reference line");
    for (int i = 0; i < n; ++i) {
        const TplLatLonState& tp = traj[i];
        const StationFrame rp = station_frame(env->ref_line, env->ref_step, tp.s);
        TplLatLonCartState cp;
        std::memset(&cp, 0, sizeof cp);
        cp.t = tp.t;
        cp.x = env->x_offset + rp.x - (double)tp.l * rp.sin_h;
        cp.y = env->y_offset + rp.y + (double)tp.l * rp.cos_h;
        if ((double)tp.ds < 1e-3) cp.heading = rp.heading;
        else cp.heading = std::atan((double)tp.dl / (double)tp.ds) + rp.heading;
        const double a = (1.0 - rp.k * (double)tp.l) * (double)tp.ds;
        cp.v = std::sqrt(a * a + (double)(tp.dl * tp.dl));
        cp.cost = tp.cost;
        cp.constr = tp.constr;
        cp.flags_constr = tp.flags_constr;
        out[i] = cp;
    }
    for (int i = 1; i < n; ++i) {
        TplLatLonCartState& cp0 = out[i - 1];
        TplLatLonCartState& cp1 = out[i];
        const float dx = (float)(cp1.x - cp0.x);
        const float dy = (float)(cp1.y - cp0.y);
        const float ds = std::sqrt(dx * dx + dy * dy);
        cp1.distance = cp0.distance + (double)ds;
        cp0.a = (cp1.v - cp0.v) / (cp1.t - cp0.t);
        if ((double)ds >= 1e-3) cp0.k = short_angle_dist(cp0.heading, cp1.heading) / (double)ds;
        else cp1.k = 0.0;
        cp1.k = cp0.k;
        cp1.a = cp0.a;
    }
    return TPLHIP_OK;
}

int tplhip_lat_lon_reeval_traj(tplhip_lat_lon* h, tplhip_dp_env* env, const TplLatLonState* traj, int n, TplLatLonState* out)
{
    int rc = ll_check_env(h, env);
    if (rc) return rc;
    if (!traj || !out || n < 1) return tplhip_fail(TPLHIP_EINVAL, "bad trajectory argument");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    if ((rc = ll_upload_ref(h, env))) return rc;
    if ((rc = ll_ensure_traj(h, (size_t)std::max(n, h->params.t_steps)))) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_traj2, traj, sizeof(TplLatLonState) * (size_t)n, hipMemcpyHostToDevice, h->stream),
            "copying reeval trajectory to gpu failed");
    hipLaunchKernelGGL(k_ll_reeval, dim3((n + 63) / 64), dim3(64), 0, h->stream,
                       h->d_traj, h->d_traj2, n, h->d_ref, (int)env->ref_line.size(), env->ref_step,
                       env->d_dist_lon, h->params, env->params, h->rule);
    HIP_TRY(hipGetLastError(), "reeval trajectory failed");
    HIP_TRY(hipMemcpyAsync(out, h->d_traj, sizeof(TplLatLonState) * (size_t)n, hipMemcpyDeviceToHost, h->stream),
            "copying reeval trajectory to cpu failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "reeval trajectory failed");
    return TPLHIP_OK;
}

int tplhip_lat_lon_copy_backward_pass_tex(tplhip_lat_lon* h, int idx_t, float* out)
{
    if (!h || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!h->have_params || !h->d_tables) return tplhip_fail(TPLHIP_ESTATE, "no value tables allocated");
    if (idx_t < 0 || idx_t >= h->params.t_steps) return tplhip_fail(TPLHIP_EINVAL, "stage %d out of range", idx_t);
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    return tplhip_copy_to_host(out, h->d_tables + (size_t)idx_t * h->table_cells, sizeof(float4) * h->table_cells, h->stream,
                               "copying backward pass texture gpu to cpu failed");
}

int tplhip_lat_lon_query_backward_pass_tex(tplhip_lat_lon* h, int idx_t, float s, float ds, float l, float* out4)
{
    if (!h || !out4) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!h->have_params || !h->d_tables) return tplhip_fail(TPLHIP_ESTATE, "no value tables allocated");
    if (idx_t < 0 || idx_t >= h->params.t_steps) return tplhip_fail(TPLHIP_EINVAL, "stage %d out of range", idx_t);
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    float ss, sd, sl;
    ll_steps(h->params, &ss, &sd, &sl);
    hipLaunchKernelGGL(k_ll_query, dim3(1), dim3(1), 0, h->stream, h->d_q, h->d_tables + (size_t)idx_t * h->table_cells,
                       s, ds, l, h->params, ss, sd, sl);
    HIP_TRY(hipMemcpyAsync(out4, h->d_q, sizeof(float4), hipMemcpyDeviceToHost, h->stream), "querying backward pass texture failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "querying backward pass texture failed");
    return TPLHIP_OK;
}

} // extern "C"
