// dp_lon_kernels.h -- device code of the DP (s,v,a) planner's general path (included by dp_lon.hip only, which holds
// the host driver): action descriptors, stage prelude, the general backward sweep (two-pass argmin, IEEE divisions, any
// input), the division proof, the forward rollout with one step of speculation, re-evaluation, the path distance map and
// the best-plan reduction.  The plain-fp32 sweeps that take over wherever their preconditions hold are in dp_sweep.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#include "dp_lon_internal.h"
#include "dp_env_internal.h"

using namespace tplhip;

namespace {

// ------------------------------------------------------------------------
// descriptor table: one thread per (row, action)
// ------------------------------------------------------------------------
__device__ __forceinline__ ActDesc make_action_desc(const TplDpLonParams& p, FilterRule rule, int ia, int iv, int k)
{
    TplLonState tp;
    tp.t = 0.0f; tp.s = 0.0f; tp.j = 0.0f; tp.cost = 0.0f; tp.constr = 0.0f;
    tp.v = grid_pt(p.v_min, (float)iv, p._v_step_size, rule);      // lon_planner.cu:198
    tp.a = grid_pt(p.a_min, (float)ia, p._a_step_size, rule);      // lon_planner.cu:199
    const float dt = p.dt;
    const float j = action_jerk(p, k, N_BWD, rule);
    const TplLonState tn = lon_dynamics(tp, j, dt, rule);

    ActDesc d;
    if (rule.fma) {
        // contracted station update (rule R1 = 1): s' = fma(j_term, dt, fma(a_term, dt, (double)fmaf(vdt, dt, s))) -- the
        // last factor dt of each product is fused with its sum, so the descriptor carries the factors in front of it
        d.a_term = (0.5 * (double)tp.a) * (double)dt;
        d.j_term = (((1.0 / 6.0) * (double)j) * (double)dt) * (double)dt;
        d.vdt = tp.v;
    } else {
        d.a_term = ((0.5 * (double)tp.a) * (double)dt) * (double)dt;
        d.j_term = ((((1.0 / 6.0) * (double)j) * (double)dt) * (double)dt) * (double)dt;
        d.vdt = tp.v * dt;
    }
    d.j = j;
    d.wj_cost = state_action_cost(p, j, dt);
    // lon_planner.cu:125-130: a' < a_min and a' > a_max exclude each other, and adding the
    // +0 of the branch not taken leaves a float unchanged, so one addend carries both tests
    d.a_viol = (tn.a < p.a_min) ? (p.a_min - tn.a) : ((tn.a > p.a_max) ? (tn.a - p.a_max) : 0.0f);
    const Tap tv = make_tap(node_coord(tn.v, p.v_min, p.v_max, p.v_steps), p.v_steps, rule);
    const Tap ta = make_tap(node_coord(tn.a, p.a_min, p.a_max, p.a_steps), p.a_steps, rule);
    d.wa = ta.w; d.wv = tv.w;
    d.iwa = 1.0f - ta.w; d.iwv = 1.0f - tv.w;
    d.tn_v = tn.v;
    d.r00 = ta.i0 * p.v_steps + tv.i0;
    d.r10 = ta.i1 * p.v_steps + tv.i0;
    d.r01 = ta.i0 * p.v_steps + tv.i1;
    d.r11 = ta.i1 * p.v_steps + tv.i1;
    d.flags = ((d.r00 == d.r10 && d.r01 == d.r11) || d.wa == 0.0f) ? ACT_SINGLE_A : 0;
    d.tn_v_d = (double)tn.v;
    return d;
}

__global__ void k_build_actions(ActDesc* __restrict__ acts, TplDpLonParams p, FilterRule rule)
{
    const int nrows = p.v_steps * p.a_steps;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * N_BWD) return;
    const int row = idx / N_BWD;
    const int k = idx - row * N_BWD;
    const int ia = row / p.v_steps;
    const int iv = row - ia * p.v_steps;
    ActDesc d = make_action_desc(p, rule, ia, iv, k);
    // ACT_SHARED_UP: the row the two-rows-per-lane sweeps pair this one with -- (v + 1, a), or this row again where v + 1 is past
    // the grid -- takes for action k, as the LOWER pair of its 2 x 2 source window, exactly this row's UPPER pair under the same
    // weight along a (the test the sweeps used to make per workgroup from both descriptors, up to 27 dependent scalar loads)
    const ActDesc y = make_action_desc(p, rule, ia, iv + 1 < p.v_steps ? iv + 1 : iv, k);
    if (d.r01 == y.r00 && d.r11 == y.r10 && __float_as_uint(d.wa) == __float_as_uint(y.wa)) d.flags |= ACT_SHARED_UP;
    acts[idx] = d;
}

// ------------------------------------------------------------------------
// per-(stage, station) prelude: interpPath(tp.s), dist map lookup, progress
// cost.  evalNode lon_planner.cu:80-86.  grid = t_steps blocks.
// ------------------------------------------------------------------------
// blockIdx.y = scenario of a batch (strides in elements between consecutive scenarios; 0 = shared).
__global__ void k_stage_prelude(float* __restrict__ pre_vmax, float* __restrict__ pre_sdist,
                                double* __restrict__ pre_prog,
                                const float* __restrict__ path, const float* __restrict__ dist_ts,
                                TplDpLonParams p, TplDpEnvParams ep,
                                size_t stride_pre, size_t stride_path, size_t stride_dist, FilterRule rule)
{
    const int idx_t = blockIdx.x;
    if (idx_t < 1) return;
    pre_vmax += blockIdx.y * stride_pre;
    pre_sdist += blockIdx.y * stride_pre;
    path += blockIdx.y * stride_path;
    dist_ts += blockIdx.y * stride_dist;
    const float t = grid_pt(p.dt_start, (float)(idx_t - 1), p.dt, rule);       // lon_planner.cu:196
    for (int is = threadIdx.x; is < p.s_steps; is += blockDim.x) {
        const float s = grid_pt(p.s_min, (float)is, p._s_step_size, rule);     // lon_planner.cu:197
        const PathTap pt = path_tap(s, p.path_step_size, p.path_steps);
        const float cps_s = path_lerp(pt, path[pt.i0 * 7 + 2], path[pt.i1 * 7 + 2], rule);
        const float v_max = path_lerp(pt, path[pt.i0 * 7 + 5], path[pt.i1 * 7 + 5], rule);
        const float s_dist = s_dist_of(interp_dist_map_path(ep, dist_ts, t, cps_s, rule), p, rule);
        pre_vmax[(size_t)idx_t * p.s_steps + is] = v_max;
        pre_sdist[(size_t)idx_t * p.s_steps + is] = s_dist;
        if (idx_t == 1 && blockIdx.y == 0) {
            // (contracted arithmetic fuses the product with the cost sum: the table then carries the multiplicand)
            pre_prog[is] = rule.fma ? fabs(1000.0 - (double)s) : (double)p.w_progress * fabs(1000.0 - (double)s);
        }
    }
}

// ------------------------------------------------------------------------
// backward stage sweep (lonBackwardStep, lon_planner.cu:180-209)
//
// blockDim.x = s_pad * rows_per_block, s_pad = s_steps rounded up to 64, so a
// wavefront never straddles two rows and the row index is wave-uniform.
// KCHUNK actions are blended into LDS per barrier.
// LDS: float4 R[KCHUNK][blockDim] | float path_vmax[path_steps]
// ------------------------------------------------------------------------
// This is the GENERAL sweep (any input: two-pass argmin, IEEE divisions); the plain-fp32 sweeps of dp_sweep.hip take
// over wherever their preconditions hold.
// ORDERED: rule.order != 0 (another blend order / the eight-corner sums) -- a template bit so that the default instance keeps
// its registers (a run-time test of rule.order in phase 2 cost it 20 us per 256x128x64 sweep).
template <bool TERMINAL, int KCHUNK, bool USE_BUF, bool ORDERED = false>
__global__ void k_backward(const Node* __restrict__ next, Node* __restrict__ out,
                           const ActDesc* __restrict__ acts,
                           const float* __restrict__ path,
                           const float* __restrict__ pre_vmax,    // this stage, [s_steps]
                           const float* __restrict__ pre_sdist,   // this stage, [s_steps]
                           const double* __restrict__ pre_prog,   // [s_steps]
                           TplDpLonParams p, int s_pad, int rows_per_block, int nblocks,
                           int tile_v, int tile_a,
                           size_t stride_tables, size_t stride_pre, size_t stride_path, FilterRule rule)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // scenario batch: blockIdx.y selects the scenario's tables, stage prelude and path (strides 0 = shared)
    next += blockIdx.y * stride_tables;
    out += blockIdx.y * stride_tables;
    pre_vmax += blockIdx.y * stride_pre;
    pre_sdist += blockIdx.y * stride_pre;
    path += blockIdx.y * stride_path;

    // Traversal order (speed only; any order gives the same table).  Blocks b and b+8 share an
    // XCD and its 4 MiB L2 (round-robin dispatch).  Action k of row (v,a) reads next-stage rows
    // around (v + a dt + j_k dt^2/2, a + j_k dt): the a' windows of the nine actions are ~A/7 rows
    // apart and slide by ONE row per a-step, the v' window by a fraction of a row.  So
    //  * each XCD owns a contiguous range of v indices,
    //  * inside it, strips of `tile_v` v-indices are walked from a = 0 upwards, a-tiles of
    //    `tile_a` rows at a time (v fastest inside a tile):
    // the rows one a-tile fetched for action k+1 are the rows the next a-tile needs for action
    // k, and a strip's working set (9 windows x (tile_a+1) a-rows x (tile_v+3) v-rows) stays
    // inside the L2.  The first version walked v fastest across the whole v axis: 440 MB of L2
    // misses per 256x128x64 sweep against 33.5 MB of table.
    const int S = p.s_steps, V = p.v_steps, A = p.a_steps;
    const int nrows = V * A;
    const int tid = threadIdx.x;
    const int rib = tid / s_pad;
    const int is = tid - rib * s_pad;
    const int xcd = blockIdx.x & 7;
    const int v_lo = (xcd * V) >> 3, v_hi = ((xcd + 1) * V) >> 3;      // this XCD's v range
    const int vx = v_hi - v_lo;
    const int q = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3) * rows_per_block + rib);   // position in the XCD's order
    const bool row_ok = q < vx * A;
    int ia = 0, iv = 0;
    if (row_ok) {
        const int full = vx / tile_v;                                   // strips of full width
        int strip = q / (A * tile_v), w = tile_v, rem = q - strip * (A * tile_v);
        if (strip >= full) { strip = full; w = vx - full * tile_v; rem = q - full * (A * tile_v); }
        const int tile = rem / (tile_a * w);
        const int r = rem - tile * (tile_a * w);
        ia = tile * tile_a + r / w;
        iv = v_lo + strip * tile_v + (r - (r / w) * w);
    }
    const int row = ia * V + iv;
    const bool active = row_ok && is < S;
    (void)nblocks;

    const float s = grid_pt(p.s_min, (float)is, p._s_step_size, rule);
    const float v = grid_pt(p.v_min, (float)iv, p._v_step_size, rule);
    const float a = grid_pt(p.a_min, (float)ia, p._a_step_size, rule);

    float v_max = 0.0f, s_dist = 0.0f;
    double prog = 0.0;
    if (active) {
        v_max = pre_vmax[is];
        s_dist = pre_sdist[is];
        prog = pre_prog[is];
    }

    // stateCost lon_planner.cu:40-55 (progress term precomputed per station)
    float sc = 0.0f;
    sc += p.w_a * (a * a);
    sc = rule.fma ? (float)fma((double)p.w_progress, prog, (double)sc) : (float)((double)sc + prog);
    {
        const float gap = madf(v, p.time_gap, p.gap_min, rule) - s_dist;
        const double m = fmax(0.0, (double)gap);
        sc = (float)madd((double)p.w_safety_dist, m, (double)sc, rule);
    }
    float sg = 0.0f;
    if (v > v_max) sg += v - v_max;

    if (TERMINAL) {
        // lon_planner.cu:93-97: cost only; constr and j keep their zero init
        if (active) st_node(out + (size_t)row * S + is, sc, 0.0f, 0.0f);
        return;
    }

    float4* R = reinterpret_cast<float4*>(smem);
    float* pv = reinterpret_cast<float*>(smem + (size_t)KCHUNK * blockDim.x * sizeof(float4));
    // (USE_BUF: the host checked that one stage table is addressable with 32-bit byte offsets)
    const __amdgpu_buffer_rsrc_t next_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<Node*>(next), 0, USE_BUF ? (int)((size_t)nrows * S * sizeof(Node)) : 0, 0x00020000);
    for (int i = tid; i <= p.path_steps; i += blockDim.x) pv[i] = path[min(i, p.path_steps - 1) * 7 + 5];   // last sample twice

    const float s_range = p.s_max - p.s_min;
    const float s_scale = (float)(S - 1);
    float res_c[N_BWD], res_g[N_BWD];

#pragma unroll
    for (int k0 = 0; k0 < N_BWD; k0 += KCHUNK) {
        if (k0 > 0) __syncthreads();       // everyone is done reading R of the last chunk
        // ---- phase 1: blend the four neighbouring next-stage rows (a, then v)
#pragma unroll
        for (int kk = 0; kk < KCHUNK; ++kk) {
            const int k = k0 + kk;
            if (active && !ORDERED) {                          // (another blend order takes its eight taps in phase 2)
                const ActDesc* d = acts + (size_t)row * N_BWD + k;
                // (clamped successors name the same row twice; those repeats hit L1 -- branching
                // around them serialises the loads and more than doubles the sweep time)
                float4 p00, p10, p01, p11;
                if (USE_BUF) {
                    // buffer loads: 128-bit descriptor of the whole next-stage table, row base as
                    // SCALAR offset, station as one 32-bit lane offset shared by all 36 loads of the
                    // cell -- no per-load 64-bit address arithmetic on the VALU
                    const unsigned voff = (unsigned)is * 12u;
                    const unsigned row_bytes = (unsigned)S * 12u;
                    p00 = buf_load_node(next_rsrc, voff, (unsigned)d->r00 * row_bytes);
                    p10 = buf_load_node(next_rsrc, voff, (unsigned)d->r10 * row_bytes);
                    p01 = buf_load_node(next_rsrc, voff, (unsigned)d->r01 * row_bytes);
                    p11 = buf_load_node(next_rsrc, voff, (unsigned)d->r11 * row_bytes);
                } else {
                    p00 = ld_node(next + (size_t)d->r00 * S + is);
                    p10 = ld_node(next + (size_t)d->r10 * S + is);
                    p01 = ld_node(next + (size_t)d->r01 * S + is);
                    p11 = ld_node(next + (size_t)d->r11 * S + is);
                }
                const float wa = d->wa, iwa = d->iwa, wv = d->wv, iwv = d->iwv;
                float4 b0, b1, bb;
                if (rule.lerp == 0) {                  // (the complements 1 - w come with the descriptor)
                    b0 = lerp4w(p00, p10, wa, iwa);
                    b1 = lerp4w(p01, p11, wa, iwa);
                    bb = lerp4w(b0, b1, wv, iwv);
                } else {
                    b0 = lerp4(p00, p10, wa, rule);
                    b1 = lerp4(p01, p11, wa, rule);
                    bb = lerp4(b0, b1, wv, rule);
                }
                R[kk * blockDim.x + tid] = bb;
            }
        }
        __syncthreads();
        // ---- phase 2: per-cell successor station, 1-D lerp, costs
#pragma unroll
        for (int kk = 0; kk < KCHUNK; ++kk) {
            const int k = k0 + kk;
            float c = 0.0f, g = 0.0f;
            if (active) {
                const ActDesc* d = acts + (size_t)row * N_BWD + k;
                // lonDynamics station part, lon_planner.cu:244
                float tn_s;
                if (rule.fma) {                        // see k_build_actions: the descriptor carries the factors in front of dt
                    const float sf = fmaf(d->vdt, p.dt, s);
                    const double sd = fma(d->j_term, (double)p.dt, fma(d->a_term, (double)p.dt, (double)sf));
                    tn_s = fmaxf(s, (float)sd);
                } else {
                    const float sf = s + d->vdt;
                    double sd = (double)sf + d->a_term;
                    sd = sd + d->j_term;
                    tn_s = fmaxf(s, (float)sd);
                }
                // readNode coordinate along s (lon_planner.cu:35).  tn_s >= s >= s_min, so the
                // coordinate is never negative and only the upper clamp can bite.
                const float xs = tn_s - p.s_min;
                const float qs = xs / s_range;
                const float cs = qs * s_scale + 0.5f;         // == (float)((double)c + 0.5), see node_coord
                const float xb = cs - 0.5f;
                const float fl = floorf(xb);
                const float ws = quant_weight(xb - fl, rule);
                const int i_lo = (int)fminf(fl, s_scale);
                const int i_hi = min(i_lo + 1, S - 1);
                const float4* Rk = R + kk * blockDim.x + rib * s_pad;
                float4 node;
                if (!ORDERED) {
                    node = lerp4(Rk[i_lo], Rk[i_hi], ws, rule);
                } else {
                    // rule R3 with another order (s -> v -> a) or as the eight-corner sum: no row can be blended ahead of the
                    // station, so the cell gathers its eight taps itself (neighbouring lanes read neighbouring cells: the
                    // requests coalesce, the repeats hit L1).  The price of following such a dump: DESIGN.md 0.
                    const Node* r00 = next + (size_t)d->r00 * S; const Node* r01 = next + (size_t)d->r01 * S;
                    const Node* r10 = next + (size_t)d->r10 * S; const Node* r11 = next + (size_t)d->r11 * S;
                    const float4 n8[8] = {ld_node(r00 + i_lo), ld_node(r00 + i_hi), ld_node(r01 + i_lo), ld_node(r01 + i_hi),
                                          ld_node(r10 + i_lo), ld_node(r10 + i_hi), ld_node(r11 + i_lo), ld_node(r11 + i_hi)};
                    node = blend8_nodes(n8, ws, d->wv, d->wa, rule);
                }
                // interpPath(tn.s).v_max, lon_planner.cu:114
                const float alpha = tn_s / p.path_step_size;
                const PathTap pt = path_tap_q(alpha, p.path_steps);
                const float vmn = path_lerp(pt, pv[pt.i0], pv[pt.i1], rule);

                c = sc;
                g = sg;
                c += node.x;
                g += node.y;
                c = madf(p.w_snap, (node.z - d->j) * (node.z - d->j), c, rule);
                c += d->wj_cost;
                // lon_planner.cu:114-118 compares and subtracts in double; float -> double is
                // exact and order preserving, so the test itself can stay in float
                if (d->tn_v > vmn) g = (float)((double)g + (d->tn_v_d - (double)vmn));
                const float s_change = tn_s - s;
                if (s_change > s_dist) {
                    g += s_change - s_dist;
                }
                g += d->a_viol;
            }
            res_c[k] = c;
            res_g[k] = g;
        }
    }

    if (!active) return;

    // lexicographic argmin, lon_planner.cu:153-170
    float constr_min = INFINITY;
#pragma unroll
    for (int k = 0; k < N_BWD; ++k) {
        if (res_g[k] < constr_min) constr_min = res_g[k];
    }
    float cost_min = INFINITY;
    int idx_min = 0;
#pragma unroll
    for (int k = 0; k < N_BWD; ++k) {
        if (res_g[k] > constr_min) continue;
        if (res_c[k] < cost_min) {
            cost_min = res_c[k];
            idx_min = k;
        }
    }
    const float j_best = action_jerk(p, idx_min, N_BWD, rule);
    st_node(out + (size_t)row * S + is, cost_min, constr_min, j_best);
}

// ------------------------------------------------------------------------
// Exhaustive check of the two cheap division forms of the plain-fp32 sweeps against the IEEE quotient for EVERY float
// in [0, x_max] (all are odd-symmetric, so the sign is covered): ~1e9 candidates, about a millisecond, once per divisor.
//  * fma(x, r_hi, x*r_lo) with r_hi + r_lo = 1/c to ~49 bits: the sum is within ~2^-47 of x/c before its one rounding,
//    two plain fp32 instructions -- exact where no quotient of the range comes that close to a rounding boundary,
//    which only the brute force can tell;
//  * x * r_hi alone: exact for a power-of-two divisor.
// A form is only used if its flag stays 0.
// scale > 0 (the station-range divisor): the forms are compared on what the sweep derives from the quotient -- the
// filter coordinate ((q * scale + 0.5) - 0.5, lon_planner.cu:35 + arr_tex_surf.cuh:142-149) -- not on the quotient
// itself: for dividends below ~1e-34 their x * r_lo term underflows and the quotient's last bit can differ, but every
// quotient below 2^-26 / scale yields the coordinate 0 either way.  scale == 0 (the path step): raw quotients, for the
// dividends 0 and [2^-96, max]: below 2^-96 the products of the reciprocal forms underflow and their last bit is not the
// division's for ANY divisor that is not a power of two (brute force on the host: every mismatch of 0.1, 0.2, 0.3, 0.7,
// 0.75, 1.3 lies below 1.3e-31) -- the sweeps send a wavefront that holds such a dividend through an IEEE division
// (kDivTinyBits, dp_sweep.hip), so the proof does not have to cover them.
// ------------------------------------------------------------------------
constexpr unsigned int kDivTinyBits = 0x0F800000u;      // 2^-96 as a float

__global__ void k_verify_div(unsigned int* __restrict__ flags, float c, float r, float r_lo, float scale, unsigned int max_bits)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    bool bad_2 = false, bad_p = false, bad_n = false;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= max_bits; b += stride) {
        if (scale == 0.0f && b > 0 && b < kDivTinyBits) continue;
        const float x = __uint_as_float((unsigned int)b);
        const float qf = x / c;
        const float q2 = fmaf(x, r, x * r_lo);                       // two-float reciprocal r + r_lo
        const float qp = x * r;                                      // exact for a power-of-two divisor
        const float qn = fmaf(fmaf(-qp, c, x), r, qp);               // one Newton step on the quotient: residual, correction
        if (scale > 0.0f) {
            const unsigned int want = __float_as_uint((qf * scale + 0.5f) - 0.5f);
            bad_2 |= (want != __float_as_uint((q2 * scale + 0.5f) - 0.5f));
            bad_p |= (want != __float_as_uint((qp * scale + 0.5f) - 0.5f));
            bad_n |= (want != __float_as_uint((qn * scale + 0.5f) - 0.5f));
        } else {
            bad_2 |= (__float_as_uint(qf) != __float_as_uint(q2));
            bad_p |= (__float_as_uint(qf) != __float_as_uint(qp));
            bad_n |= (__float_as_uint(qf) != __float_as_uint(qn));
        }
    }
    if (bad_2) atomicOr(&flags[0], 1u);      // two-float reciprocal is not exact here
    if (bad_p) atomicOr(&flags[1], 1u);      // a bare multiplication by the reciprocal is not exact here
    if (bad_n) atomicOr(&flags[2], 1u);      // reciprocal + one Newton step is not exact here
}

// ------------------------------------------------------------------------
// evalNode at an arbitrary trajectory point, one wavefront, action k on lane k.
// All lanes hold the same tp.  lon_planner.cu:72-178
// ------------------------------------------------------------------------
// Minimum over the 64 lanes with data-parallel-primitive moves (no LDS round trip: __shfl_xor is a ds_bpermute_b32 of
// ~100 cycles on the dependent chain of a lone wavefront, six of them per reduction, two reductions per rollout step):
// inside each row of 16 lanes by quad swaps and mirrors, then row -> row with the two row broadcasts; the result forms
// in lane 63.  Inputs are never NaN (callers map NaN to +inf), so the order of the comparisons does not matter.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_min(float x)
{
    x = fminf(x, dpp_move<0xB1, 0xf>(x));       // quad_perm [1, 0, 3, 2]
    x = fminf(x, dpp_move<0x4E, 0xf>(x));       // quad_perm [2, 3, 0, 1]
    x = fminf(x, dpp_move<0x141, 0xf>(x));      // row_half_mirror
    x = fminf(x, dpp_move<0x140, 0xf>(x));      // row_mirror: every lane of a row holds the row's minimum
    x = fminf(x, dpp_move<0x142, 0xa>(x));      // row_bcast:15 into rows 1 and 3
    x = fminf(x, dpp_move<0x143, 0xc>(x));      // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// One lane's share of evalNode (lon_planner.cu:72-137): the state terms at tp and, unless the node is terminal, the
// action with jerk j.  All lanes of a candidate set hold the same tp.
struct LaneEval { float cost, constr, sc; bool terminal; };

__device__ __forceinline__ LaneEval eval_lane(const TplLonState& tp, float j, const Node* __restrict__ table_next,
                                              const float* __restrict__ path, const float* __restrict__ dist_ts,
                                              const TplDpLonParams& p, const TplDpEnvParams& ep, float dt, FilterRule rule)
{
    const PathTap pt = path_tap(tp.s, p.path_step_size, p.path_steps);
    const float cps_s = path_lerp(pt, path[pt.i0 * 7 + 2], path[pt.i1 * 7 + 2], rule);
    const float v_max = path_lerp(pt, path[pt.i0 * 7 + 5], path[pt.i1 * 7 + 5], rule);
    const float s_dist = s_dist_of(interp_dist_map_path(ep, dist_ts, tp.t, cps_s, rule), p, rule);
    const float sc = state_cost(p, tp.s, tp.v, tp.a, s_dist, rule);
    float sg = 0.0f;
    if (tp.v > v_max) sg += tp.v - v_max;
    if (roundf(tp.t / p.dt) == (float)(p.t_steps - 1)) return LaneEval{sc, 0.0f, sc, true};    // lon_planner.cu:93-97
    const TplLonState tn = lon_dynamics(tp, j, dt, rule);
    const float4 node = read_node(p, table_next, tn.s, tn.v, tn.a, rule);
    const PathTap pn = path_tap(tn.s, p.path_step_size, p.path_steps);
    const float vmn = path_lerp(pn, path[pn.i0 * 7 + 5], path[pn.i1 * 7 + 5], rule);
    const ActionResult r = eval_action_tail(p, sc, sg, node, j, state_action_cost(p, j, dt),
                                            tn.s, tn.v, tn.a, tp.s, s_dist, vmn, rule);
    return LaneEval{r.cost, r.constr, sc, false};
}

// The lexicographic argmin of lon_planner.cu:153-170 over the lanes with `valid` set: lane of the winner (the lowest
// lane when nothing qualifies: index 0 of the reference's loop), its cost, the smallest constraint value.
struct Choice { int lane; float cost, constr; };

__device__ __forceinline__ Choice choose_action_wave(bool valid, float cost, float constr, int first_lane)
{
    // constr_min: NaN never lowers the minimum (lon_planner.cu:155)
    const float g_in = (valid && !(constr != constr)) ? constr : INFINITY;
    const float constr_min = wave_min(g_in);
    // candidates: not (constr > constr_min)  -- NaN constr stays a candidate (:163)
    const bool cand = valid && !(constr > constr_min);
    // first strictly smallest cost among candidates (:166); NaN / +inf cost never selected
    const float c_in = (cand && cost < INFINITY) ? cost : INFINITY;
    const float cost_min = wave_min(c_in);
    const unsigned long long hit = __ballot(cand && c_in == cost_min && cost_min < INFINITY);
    return Choice{hit ? (int)__builtin_ctzll(hit) : first_lane, cost_min, constr_min};
}

// evalNode with the N candidate jerks on lanes 0..N-1 of one wavefront.  Returns the chosen action, -1 for a terminal
// node (cost only; jerk and constraint keep their incoming values) and for choose_action == false (re-costs tp.j).
template <int N>
__device__ __forceinline__ int eval_node_wave(TplLonState& tp, const Node* __restrict__ table_next,
                                              const float* __restrict__ path,
                                              const float* __restrict__ dist_ts,
                                              const TplDpLonParams& p, const TplDpEnvParams& ep,
                                              float dt, bool choose_action, FilterRule rule = kDefaultRule)
{
    const int lane = threadIdx.x & 63;
    const int k = choose_action ? lane : 0;
    const float j = choose_action ? action_jerk(p, k < N ? k : N - 1, N, rule) : tp.j;
    const LaneEval e = eval_lane(tp, j, table_next, path, dist_ts, p, ep, dt, rule);
    if (e.terminal) {
        tp.cost = e.sc;
        return -1;
    }
    if (!choose_action) {
        tp.cost = e.cost;
        tp.constr = e.constr;
        return -1;
    }
    const Choice c = choose_action_wave(lane < N, e.cost, e.constr, 0);
    tp.j = action_jerk(p, c.lane, N, rule);
    tp.cost = c.cost;
    tp.constr = c.constr;
    return c.lane;
}

// lonForwardStep chain, lon_planner.cu:211-225 + :365-386.
// A step is a chain of dependent look-ups (path -> distance map -> next-stage table), and loads and
// stores of a wavefront complete in order: the trajectory therefore collects in an LDS ring that
// leaves in one coalesced write every FWD_RING steps (a store per step would put a write
// acknowledgement in front of the next step's first load), and where they fit (IN_LDS) the path and
// the distance map are copied to LDS once, so only the table reads of a step go to memory.
constexpr int FWD_RING = 16;

// Scenario batches: rollout r belongs to scenario r (one rollout each); its tables, path and distance map are
// stride_* elements further on (0 = shared by all rollouts, the single-plan case).
// RULE_RT: the filter rule is the kernel argument (a non-default rule was set); otherwise the default folds away.
// evalNode<N> as ONE function for the sequential rollout below (the same arithmetic as eval_lane + choose_action_wave, kept
// in one piece: split that way and inlined, the one-wavefront rollout of a 256x128x64 x 50 plan measured 143 us instead of
// 118 -- the compiler schedules the eight table reads of a step differently; profiles/r03_dp_forward_speculation.txt)
template <int N>
__device__ __forceinline__ void eval_node_wave_seq(TplLonState& tp, const Node* __restrict__ table_next,
                                                   const float* __restrict__ path, const float* __restrict__ dist_ts,
                                                   const TplDpLonParams& p, const TplDpEnvParams& ep, float dt, FilterRule rule)
{
    const int lane = threadIdx.x & 63;
    const PathTap pt = path_tap(tp.s, p.path_step_size, p.path_steps);
    const float cps_s = path_lerp(pt, path[pt.i0 * 7 + 2], path[pt.i1 * 7 + 2], rule);
    const float v_max = path_lerp(pt, path[pt.i0 * 7 + 5], path[pt.i1 * 7 + 5], rule);
    const float s_dist = s_dist_of(interp_dist_map_path(ep, dist_ts, tp.t, cps_s, rule), p, rule);
    const float sc = state_cost(p, tp.s, tp.v, tp.a, s_dist, rule);
    float sg = 0.0f;
    if (tp.v > v_max) sg += tp.v - v_max;
    if (roundf(tp.t / p.dt) == (float)(p.t_steps - 1)) {    // lon_planner.cu:93-97
        tp.cost = sc;
        return;
    }
    const float j = action_jerk(p, lane < N ? lane : N - 1, N, rule);
    const TplLonState tn = lon_dynamics(tp, j, dt, rule);
    const float4 node = read_node(p, table_next, tn.s, tn.v, tn.a, rule);
    const PathTap pn = path_tap(tn.s, p.path_step_size, p.path_steps);
    const float vmn = path_lerp(pn, path[pn.i0 * 7 + 5], path[pn.i1 * 7 + 5], rule);
    const ActionResult r = eval_action_tail(p, sc, sg, node, j, state_action_cost(p, j, dt),
                                            tn.s, tn.v, tn.a, tp.s, s_dist, vmn, rule);
    const bool valid = lane < N;
    const float g_in = (valid && !(r.constr != r.constr)) ? r.constr : INFINITY;       // lon_planner.cu:155
    const float constr_min = wave_min(g_in);
    const bool cand = valid && !(r.constr > constr_min);                              // :163
    const float c_in = (cand && r.cost < INFINITY) ? r.cost : INFINITY;               // :166
    const float cost_min = wave_min(c_in);
    const unsigned long long hit = __ballot(cand && c_in == cost_min && cost_min < INFINITY);
    const int idx_min = hit ? (int)__builtin_ctzll(hit) : 0;
    tp.j = action_jerk(p, idx_min, N, rule);
    tp.cost = cost_min;
    tp.constr = constr_min;
}

// The rollout, one wavefront and one step per round trip: the form for table sets beyond the Infinity Cache, where
// speculation loses (see k_forward below).  Arguments as k_forward.
template <bool IN_LDS, bool RULE_RT>
__global__ void __launch_bounds__(64) k_forward_seq(TplLonState* __restrict__ trajs, const TplLonState* __restrict__ inits,
                                                    const Node* __restrict__ tables, size_t table_cells,
                                                    const float* __restrict__ path, const float* __restrict__ dist_ts,
                                                    TplDpLonParams p, TplDpEnvParams ep,
                                                    size_t stride_tables, size_t stride_path, size_t stride_dist, FilterRule rule_arg,
                                                    TplLonState* __restrict__ mirror, unsigned int* __restrict__ done_flag, unsigned int done_seq)
{
    const FilterRule rule = RULE_RT ? rule_arg : kDefaultRule;
    __builtin_amdgcn_s_setprio(3);
    tables += blockIdx.x * stride_tables;
    path += blockIdx.x * stride_path;
    dist_ts += blockIdx.x * stride_dist;
    extern __shared__ __attribute__((aligned(16))) char fwd_smem[];
    float* ring = reinterpret_cast<float*>(fwd_smem);                       // FWD_RING x LonState (7 floats)
    float* l_path = ring + FWD_RING * 7;
    float* l_dist = l_path + p.path_steps * 7;
    const int lane = threadIdx.x;
    if constexpr (IN_LDS) {
        for (int e = lane; e < p.path_steps * 7; e += 64) l_path[e] = path[e];
        for (int e = lane; e < ep.t_steps * ep.s_steps; e += 64) l_dist[e] = dist_ts[e];
        __syncthreads();
    }
    const float* pth = IN_LDS ? l_path : path;
    const float* dst = IN_LDS ? l_dist : dist_ts;
    const int r = blockIdx.x;
    const int T = p.t_steps;
    float* out = reinterpret_cast<float*>(trajs + (size_t)r * T);
    float* out2 = mirror ? reinterpret_cast<float*>(mirror + (size_t)r * T) : nullptr;
    TplLonState tp = inits[r];
    for (int i = 0; i < T; ++i) {
        const float dt = (i == 0) ? p.dt_start : p.dt;
        eval_node_wave_seq<N_FWD>(tp, tables + (size_t)min(T - 1, i + 1) * table_cells, pth, dst, p, ep, dt, rule);
        const int slot = i % FWD_RING;
        if (lane == 0) {
            float* q = ring + slot * 7;
            q[0] = tp.t; q[1] = tp.s; q[2] = tp.v; q[3] = tp.a; q[4] = tp.j; q[5] = tp.cost; q[6] = tp.constr;
        }
        if (slot == FWD_RING - 1 || i == T - 1) {
            __syncthreads();
            const int n = (slot + 1) * 7, base = (i - slot) * 7;
            for (int e = lane; e < n; e += 64) {
                const float x = ring[e];
                out[base + e] = x;
                if (out2) out2[base + e] = x;
            }
            __syncthreads();
        }
        if (i < T - 1) tp = lon_dynamics(tp, tp.j, dt, rule);
    }
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (lane == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------
// The rollout with ONE STEP OF SPECULATION: a step cannot start before the step before it has chosen its action,
// but that action is one of 21 -- so while one wavefront evaluates step i (evalNode<21> at the known state), seven
// more evaluate step i+1 at ALL 21 states the choice can lead to (21 x 21 candidate actions, three states of 21 lanes
// per wavefront).  When the choice of step i is known, the 21 results behind it are one masked argmin away: two steps
// per round trip through path, distance map and value table instead of one.  Every candidate is evaluated by exactly
// the arithmetic of eval_lane / choose_action_wave, so the trajectory is that of the sequential loop, bit for bit
// (one wavefront per rollout and one step per round trip measured 18.5 us for the 10 steps of the default grid and
// 2.9 us per step on the 256x128x64 tables; this form 16.2 us: profiles/r03_dp_forward_speculation.txt).
// A terminal node at step i chooses nothing (its successor follows the incoming jerk, which is not among the
// speculated states in general): the wavefront of step i then evaluates step i+1 by itself.
// blockDim = 512: wavefronts 0..6 speculate, wavefront 7 evaluates the known state.
// For table sets beyond the Infinity Cache the host launches k_forward_seq instead: there a round waits for the slowest
// of 441 x 8 scattered HBM reads instead of 21 x 8, and two steps in 6.9 us lose against 2 x 2.4 us.
// ------------------------------------------------------------------------
template <bool IN_LDS, bool RULE_RT>
__global__ void __launch_bounds__(512) k_forward(TplLonState* __restrict__ trajs, const TplLonState* __restrict__ inits,
                                                      const Node* __restrict__ tables, size_t table_cells,
                                                      const float* __restrict__ path, const float* __restrict__ dist_ts,
                                                      TplDpLonParams p, TplDpEnvParams ep,
                                                      size_t stride_tables, size_t stride_path, size_t stride_dist, FilterRule rule_arg,
                                                      TplLonState* __restrict__ mirror, unsigned int* __restrict__ done_flag, unsigned int done_seq)
{
    // mirror (may be null): page-locked HOST memory that receives the trajectories too -- the synchronous update() reads
    // its result there when the stream has drained, with no copy command behind the rollout
    const FilterRule rule = RULE_RT ? rule_arg : kDefaultRule;
    // a rollout is a handful of wavefronts on a dependent chain; in the back-to-back mode it shares its SIMDs with eight
    // sweep wavefronts of the next plan each: highest priority, so that its few instructions issue when they are ready
    __builtin_amdgcn_s_setprio(3);
    tables += blockIdx.x * stride_tables;
    path += blockIdx.x * stride_path;
    dist_ts += blockIdx.x * stride_dist;
    extern __shared__ __attribute__((aligned(16))) char fwd_smem[];
    float* ring = reinterpret_cast<float*>(fwd_smem);                       // FWD_RING x LonState (7 floats)
    float* l_path = ring + FWD_RING * 7;
    float* l_dist = l_path + p.path_steps * 7;
    // hand-over between the wavefronts, double-buffered by the parity of the round
    __shared__ TplLonState x_cur[2], x_mid[2], x_nxt[2];
    __shared__ int x_k[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if constexpr (IN_LDS) {
        for (int e = tid; e < p.path_steps * 7; e += blockDim.x) l_path[e] = path[e];
        for (int e = tid; e < ep.t_steps * ep.s_steps; e += blockDim.x) l_dist[e] = dist_ts[e];
        __syncthreads();
    }
    const float* pth = IN_LDS ? l_path : path;
    const float* dst = IN_LDS ? l_dist : dist_ts;
    const int r = blockIdx.x;
    const int T = p.t_steps;
    float* out = reinterpret_cast<float*>(trajs + (size_t)r * T);
    float* out2 = mirror ? reinterpret_cast<float*>(mirror + (size_t)r * T) : nullptr;
    TplLonState tp = inits[r];
    constexpr int w_eval = 7, step = 2;
    for (int i = 0, round = 0; i < T; i += step, ++round) {
        const int par = round & 1;
        const float dt0 = (i == 0) ? p.dt_start : p.dt;
        const Node* table1 = tables + (size_t)min(T - 1, i + 1) * table_cells;
        const Node* table2 = tables + (size_t)min(T - 1, i + 2) * table_cells;
        const bool have2 = i + 1 < T;
        // speculated state of this lane's group and its evaluation (wavefronts 0..6)
        const int grp = min(lane / N_FWD, 2), m = lane - grp * N_FWD;
        const bool lane_ok = lane < 3 * N_FWD;
        TplLonState b = tp;
        LaneEval e2 = LaneEval{0.0f, 0.0f, 0.0f, false};
        if (wave == w_eval) {
            TplLonState a = tp;
            const int k = eval_node_wave<N_FWD>(a, table1, pth, dst, p, ep, dt0, true, rule);
            if (lane == 0) { x_cur[par] = a; x_k[par] = k; }
        } else if (have2) {
            b = lon_dynamics(tp, action_jerk(p, 3 * wave + grp, N_FWD, rule), dt0, rule);            // state i+1 if action 3 wave + grp wins
            e2 = eval_lane(b, action_jerk(p, min(m, N_FWD - 1), N_FWD, rule), table2, pth, dst, p, ep, p.dt, rule);
        }
        __syncthreads();
        if (have2) {
            const int ks = x_k[par];
            if (ks >= 0) {
                if (wave == ks / 3) {
                    const int g = ks - 3 * wave;
                    if (e2.terminal) {
                        b.cost = e2.sc;
                    } else {
                        const Choice c = choose_action_wave(lane_ok && grp == g, e2.cost, e2.constr, g * N_FWD);
                        b.j = action_jerk(p, c.lane - g * N_FWD, N_FWD, rule);
                        b.cost = c.cost;
                        b.constr = c.constr;
                    }
                    if (lane == g * N_FWD) {
                        x_mid[par] = b;
                        x_nxt[par] = lon_dynamics(b, b.j, p.dt, rule);
                    }
                }
            } else if (wave == w_eval) {
                // step i was terminal: no action chosen, the successor follows the incoming jerk -- not speculated
                TplLonState a = x_cur[par];
                TplLonState c = lon_dynamics(a, a.j, dt0, rule);
                eval_node_wave<N_FWD>(c, table2, pth, dst, p, ep, p.dt, true, rule);
                if (lane == 0) {
                    x_mid[par] = c;
                    x_nxt[par] = lon_dynamics(c, c.j, p.dt, rule);
                }
            }
        }
        __syncthreads();
        const int slot = i % FWD_RING;                     // FWD_RING is even: a pair never straddles the ring
        if (tid < 7) ring[slot * 7 + tid] = reinterpret_cast<const float*>(&x_cur[par])[tid];
        if (have2 && tid >= 64 && tid < 71) ring[(slot + 1) * 7 + tid - 64] = reinterpret_cast<const float*>(&x_mid[par])[tid - 64];
        const int last = have2 ? i + 1 : i;
        if ((last % FWD_RING) == FWD_RING - 1 || last == T - 1) {
            __syncthreads();
            const int n = ((last % FWD_RING) + 1) * 7, base = (last - (last % FWD_RING)) * 7;
            for (int e = tid; e < n; e += blockDim.x) {
                const float x = ring[e];
                out[base + e] = x;
                if (out2) out2[base + e] = x;
            }
            __syncthreads();
        }
        if (have2) tp = x_nxt[par];
    }
    // done_flag (host memory, single-rollout plans): the host spins on it instead of waiting for the stream to drain.
    // The flush above ended with a workgroup barrier; every lane that stored to the mirror fences its own stores to
    // system scope before the barrier below, and one lane then publishes the sequence number.
    if (done_flag) {
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// lonReevalNode, lon_planner.cu:227-236 + :439-455: nodes are independent, one thread each
__global__ void k_reeval(TplLonState* __restrict__ traj,
                         const Node* __restrict__ tables, size_t table_cells,
                         const float* __restrict__ path, const float* __restrict__ dist_ts,
                         TplDpLonParams p, TplDpEnvParams ep, FilterRule rule)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int T = p.t_steps;
    if (i >= T) return;
    TplLonState tp = traj[i];
    const float dt = (i == 0) ? p.dt_start : p.dt;
    const int nx = min(T - 1, i + 1);
    const Node* table_next = tables + (size_t)nx * table_cells;

    const PathTap pt = path_tap(tp.s, p.path_step_size, p.path_steps);
    const float cps_s = path_lerp(pt, path[pt.i0 * 7 + 2], path[pt.i1 * 7 + 2], rule);
    const float v_max = path_lerp(pt, path[pt.i0 * 7 + 5], path[pt.i1 * 7 + 5], rule);
    const float s_dist = s_dist_of(interp_dist_map_path(ep, dist_ts, tp.t, cps_s, rule), p, rule);
    const float sc = state_cost(p, tp.s, tp.v, tp.a, s_dist, rule);
    float sg = 0.0f;
    if (tp.v > v_max) sg += tp.v - v_max;
    if (roundf(tp.t / p.dt) == (float)(p.t_steps - 1)) {
        tp.cost = sc;
    } else {
        const float j = tp.j;
        const TplLonState tn = lon_dynamics(tp, j, dt, rule);
        const float4 node = read_node(p, table_next, tn.s, tn.v, tn.a, rule);
        const PathTap pn = path_tap(tn.s, p.path_step_size, p.path_steps);
        const float vmn = path_lerp(pn, path[pn.i0 * 7 + 5], path[pn.i1 * 7 + 5], rule);
        const ActionResult r = eval_action_tail(p, sc, sg, node, j, state_action_cost(p, j, dt),
                                                tn.s, tn.v, tn.a, tp.s, s_dist, vmn, rule);
        tp.cost = r.cost;
        tp.constr = r.constr;
    }
    traj[i] = tp;
}

// ------------------------------------------------------------------------
// Path + initial states from the page-locked staging buffer (read through its device mapping) to their device arrays.
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_upload_words(const unsigned* __restrict__ src, unsigned* __restrict__ dst_a, unsigned words_a,
                                                      unsigned* __restrict__ dst_b, unsigned words_b)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < words_a + words_b; i += gridDim.x * blockDim.x) {
        const unsigned v = src[i];
        if (i < words_a) dst_a[i] = v; else dst_b[i - words_a] = v;
    }
}

// ------------------------------------------------------------------------
// What a rollout of the back-to-back mode reads besides its table set -- distance map, path, initial states -- into the
// buffers of the plan's parity: three word copies in one launch (device to device).
// ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_copy_fwd_inputs(const unsigned* __restrict__ src_a, unsigned* __restrict__ dst_a, unsigned words_a,
                                                         const unsigned* __restrict__ src_b, unsigned* __restrict__ dst_b, unsigned words_b,
                                                         const unsigned* __restrict__ src_c, unsigned* __restrict__ dst_c, unsigned words_c)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < words_a + words_b + words_c; i += gridDim.x * blockDim.x) {
        if (i < words_a) dst_a[i] = src_a[i];
        else if (i < words_a + words_b) dst_b[i - words_a] = src_b[i - words_a];
        else dst_c[i - words_a - words_b] = src_c[i - words_a - words_b];
    }
}

// ------------------------------------------------------------------------
// The traversal of the two-rows-per-lane sweeps as a table (V3Map, dp_lon_internal.h), behind k_build_actions on the same
// stream: entry bx = v3_map_block(bx) and the rows of action 0 the workgroup asks for first, copied from the action table.
// ------------------------------------------------------------------------
// deps (may be null): per workgroup the row PAIRS of the next stage its actions read, as flag indices a' * pairs_per_a + v' / 2
// (SweepChain), without repeats, -1 behind the last; *deps_overflow counts workgroups with more than CHAIN_DEPS of them.
__global__ void __launch_bounds__(256) k_fill_v3_map(V3Map* __restrict__ map, int n, const ActDesc* __restrict__ acts,
                                                     int V, int A, int tile_v, int tile_a, int skew_q16,
                                                     int* __restrict__ deps, unsigned* __restrict__ deps_overflow)
{
    const int bx = blockIdx.x * blockDim.x + threadIdx.x;
    if (bx >= n) return;
    int ia = 0, iv = 0;
    V3Map m{V3_MAP_NONE, 0, 0, 0};
    int nd = 0;
    int* my = deps ? deps + (size_t)bx * CHAIN_DEPS : nullptr;
    if (v3_map_block(bx, V, A, tile_v, tile_a, skew_q16, ia, iv)) {
        const ActDesc* d0 = acts + (size_t)(ia * V + iv) * N_BWD;
        const ActDesc* d1 = iv + 1 < V ? d0 + N_BWD : d0;
        m.where = ((unsigned)ia << 16) | (unsigned)iv;
        m.r00 = d0->r00; m.r01_lo = d0->r01; m.r01_up = d1->r01;
        if (my) {
            const int NP = (V + 1) / 2;
            bool over = false;
            for (int i = 0; i < 2; ++i) {
                const ActDesc* d = i ? d1 : d0;
                for (int k = 0; k < N_BWD; ++k) {
                    const int rows[4] = {d[k].r00, d[k].r10, d[k].r01, d[k].r11};
                    for (int r : rows) {
                        const int f = (r / V) * NP + ((r % V) >> 1);
                        bool seen = false;
                        for (int q = 0; q < nd; ++q) seen = seen || my[q] == f;
                        if (seen) continue;
                        if (nd < CHAIN_DEPS) my[nd++] = f; else over = true;
                    }
                }
            }
            if (over) atomicAdd(deps_overflow, 1u);
        }
    }
    if (my) for (int q = nd; q < CHAIN_DEPS; ++q) my[q] = -1;
    map[bx] = m;
}

// ------------------------------------------------------------------------
// updateDistMapPath, env.cu:131-158: ONE time slice of the distance map, by a whole workgroup (blockDim.x a multiple of
// 64), into LDS.  occ: [l][s][t] (t fastest).
// The reference walks idx_s from s_steps-1 down with d = occ > 0 ? 0 : d + step, d starting at 10000: at station i
// that is 0 at the nearest occupied station j >= i (10000 behind the last station if there is none) plus (j - i)
// (s_steps - i) times the step, ADDED ONE AT A TIME.  The occupancy look-ups become one bit per station (a ballot per
// wavefront), every lane finds its j in the bit row and repeats exactly those additions for its own station -- the same
// floats, without one lane walking the row through LDS (201 dependent LDS round trips: 6.4 us of a 140 us plan).
// ------------------------------------------------------------------------
__device__ __forceinline__ void dist_slice_to_lds(float* __restrict__ d_sh, unsigned long long* __restrict__ occ_bits,
                                                  const float* __restrict__ occ_lst, const float* __restrict__ path,
                                                  int path_steps, float path_step_size, const TplDpEnvParams& ep, int idx_t,
                                                  FilterRule rule = kDefaultRule)
{
    const int S = ep.s_steps;
    const int words = (S + 63) >> 6;
    for (int base = 0; base < S; base += blockDim.x) {
        const int idx_s = base + threadIdx.x;
        bool occupied = false;
        if (idx_s < S) {
            const float s = grid_pt(ep.s_min, (float)idx_s, ep._s_step_size, rule);
            const PathTap pt = path_tap(s, path_step_size, path_steps);
            const float cps_s = path_lerp(pt, path[pt.i0 * 7 + 2], path[pt.i1 * 7 + 2], rule);
            const float cps_l = path_lerp(pt, path[pt.i0 * 7 + 3], path[pt.i1 * 7 + 3], rule);
            float fs = (cps_s - ep.s_min) / ep._s_step_size;
            float fl = (cps_l - ep.l_min) / ep._l_step_size;
            fs = fminf(fmaxf(fs, -1.0f), (float)ep.s_steps);
            fl = fminf(fmaxf(fl, -1.0f), (float)ep.l_steps);
            const int js = clampi((int)fs, 0, ep.s_steps - 1);
            const int jl = clampi((int)fl, 0, ep.l_steps - 1);
            occupied = occ_lst[((size_t)jl * ep.s_steps + js) * ep.t_steps + idx_t] > 0.0f;
        }
        const unsigned long long bits = __ballot(occupied);
        if ((threadIdx.x & 63) == 0 && idx_s < S) occ_bits[idx_s >> 6] = bits;
    }
    __syncthreads();
    for (int idx_s = threadIdx.x; idx_s < S; idx_s += blockDim.x) {
        int w = idx_s >> 6;
        unsigned long long bits = occ_bits[w] >> (idx_s & 63);
        int j = -1;
        if (bits) {
            j = idx_s + __builtin_ctzll(bits);
        } else {
            for (++w; w < words; ++w) {
                bits = occ_bits[w];
                if (bits) { j = (w << 6) + __builtin_ctzll(bits); break; }
            }
        }
        float d = j >= 0 ? 0.0f : 10000.0f;
        const int n = j >= 0 ? j - idx_s : S - idx_s;
        for (int k = 0; k < n; ++k) d += ep._s_step_size;
        d_sh[idx_s] = d;
    }
    __syncthreads();
}

// LDS of one slice: the distances, then the bit row
__host__ __device__ inline size_t dist_slice_lds_bytes(int env_s_steps)
{
    return sizeof(float) * (((size_t)env_s_steps + 1) & ~(size_t)1) + sizeof(unsigned long long) * (((size_t)env_s_steps + 63) >> 6);
}

// One block per time slice; dist: [t][s].
__global__ void k_dist_map_path(float* __restrict__ dist_ts, const float* __restrict__ occ_lst,
                                const float* __restrict__ path, int path_steps, float path_step_size,
                                TplDpEnvParams ep, size_t stride_dist, size_t stride_occ, size_t stride_path,
                                FilterRule rule = kDefaultRule)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dist_ts += blockIdx.y * stride_dist;          // scenario batch, see k_stage_prelude
    occ_lst += blockIdx.y * stride_occ;
    path += blockIdx.y * stride_path;
    const int idx_t = blockIdx.x;
    const int S = ep.s_steps;
    float* d_sh = reinterpret_cast<float*>(smem);
    unsigned long long* occ_bits = reinterpret_cast<unsigned long long*>(smem + sizeof(float) * (((size_t)S + 1) & ~(size_t)1));
    dist_slice_to_lds(d_sh, occ_bits, occ_lst, path, path_steps, path_step_size, ep, idx_t, rule);
    for (int idx_s = threadIdx.x; idx_s < S; idx_s += blockDim.x) {
        dist_ts[(size_t)idx_t * S + idx_s] = d_sh[idx_s];
    }
}

// ------------------------------------------------------------------------
// The front of ONE plan in one launch: distance map, stage prelude and the terminal stage (three dependent launches of
// 6.4 + 4.6 + 6.1 us on tpl's default grid, each far below the launch floor in work).  Roles by block index:
//   [0, env.t_steps)               slice b of the distance map -> dist_ts (the rollout and reeval read it)
//   [env.t_steps, +t_steps-1)      prelude of stage 1 .. t_steps-1 (k_stage_prelude's arithmetic)
//   the rest                       rows_per_block rows of the terminal table (k_backward<TERMINAL>'s arithmetic, rule R4)
// A prelude or terminal block builds the one or two distance slices ITS stage interpolates between in its own LDS
// instead of waiting for another block's: no dependency inside the launch.
// LDS: two slices (dist_slice_lds_bytes each).
// ------------------------------------------------------------------------
template <bool RULE_RT>
__global__ void __launch_bounds__(256) k_plan_front(float* __restrict__ dist_ts, const float* __restrict__ occ_lst,
                                                    const float* __restrict__ path,
                                                    float* __restrict__ pre_vmax, float* __restrict__ pre_sdist,
                                                    double* __restrict__ pre_prog, Node* __restrict__ term_out,
                                                    TplDpLonParams p, TplDpEnvParams ep, int rows_per_block, FilterRule rule_arg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FilterRule rule = RULE_RT ? rule_arg : kDefaultRule;
    const int ES = ep.s_steps, S = p.s_steps, T = p.t_steps;
    const size_t slice_b = dist_slice_lds_bytes(ES), dist_b = sizeof(float) * (((size_t)ES + 1) & ~(size_t)1);
    float* L0 = reinterpret_cast<float*>(smem);
    unsigned long long* bits0 = reinterpret_cast<unsigned long long*>(smem + dist_b);
    float* L1 = reinterpret_cast<float*>(smem + slice_b);
    unsigned long long* bits1 = reinterpret_cast<unsigned long long*>(smem + slice_b + dist_b);
    int b = blockIdx.x;
    if (b < ep.t_steps) {
        dist_slice_to_lds(L0, bits0, occ_lst, path, p.path_steps, p.path_step_size, ep, b, rule);
        for (int idx_s = threadIdx.x; idx_s < ES; idx_s += blockDim.x) dist_ts[(size_t)b * ES + idx_s] = L0[idx_s];
        return;
    }
    b -= ep.t_steps;
    const bool terminal = b >= T - 1;
    const int stage = terminal ? T - 1 : b + 1;
    const float t = grid_pt(p.dt_start, (float)(stage - 1), p.dt, rule);       // lon_planner.cu:196
    // interpDistMapPath, env.cu:254-264, on the two slices of this stage
    const Tap tt = make_tap(env_t_idx(ep, t) + 0.5f, ep.t_steps, rule);
    dist_slice_to_lds(L0, bits0, occ_lst, path, p.path_steps, p.path_step_size, ep, tt.i0, rule);
    // (a weight of exactly 0 -- every stage time the reference produces: t_idx is a whole number -- makes the blend
    // fma(q, 0, p * 1) = p for the finite distances of a slice, whatever q: the second slice is not built then)
    const float* R1 = L0;
    if (tt.i1 != tt.i0 && tt.w != 0.0f) {
        dist_slice_to_lds(L1, bits1, occ_lst, path, p.path_steps, p.path_step_size, ep, tt.i1, rule);
        R1 = L1;
    }
    const int row0 = (b - (T - 1)) * rows_per_block, nrows = p.v_steps * p.a_steps;
    for (int is = threadIdx.x; is < S; is += blockDim.x) {
        const float s = grid_pt(p.s_min, (float)is, p._s_step_size, rule);     // lon_planner.cu:197
        const PathTap pt = path_tap(s, p.path_step_size, p.path_steps);
        const float cps_s = path_lerp(pt, path[pt.i0 * 7 + 2], path[pt.i1 * 7 + 2], rule);
        const float scq = (cps_s - ep.s_min) / (ep.s_max - ep.s_min) * (float)(ep.s_steps - 1);
        const Tap ts = make_tap(scq + 0.5f, ep.s_steps, rule);
        const float s_dist = s_dist_of(dist_blend(L0[ts.i0], R1[ts.i0], L0[ts.i1], R1[ts.i1], tt.w, ts.w, rule), p, rule);
        const double away = fabs(1000.0 - (double)s);
        const double prog = rule.fma ? away : (double)p.w_progress * away;       // (contracted: the multiplicand, see k_stage_prelude)
        if (!terminal) {
            pre_vmax[(size_t)stage * S + is] = path_lerp(pt, path[pt.i0 * 7 + 5], path[pt.i1 * 7 + 5], rule);
            pre_sdist[(size_t)stage * S + is] = s_dist;
            if (stage == 1) pre_prog[is] = prog;
            continue;
        }
        // terminal stage, lon_planner.cu:93-97: the state cost only; constr and j keep their zero init
        for (int r = row0; r < min(row0 + rows_per_block, nrows); ++r) {
            const int ia = r / p.v_steps, iv = r - ia * p.v_steps;
            const float v = grid_pt(p.v_min, (float)iv, p._v_step_size, rule);
            const float a = grid_pt(p.a_min, (float)ia, p._a_step_size, rule);
            float sc = 0.0f;                                      // stateCost lon_planner.cu:40-55
            sc += p.w_a * (a * a);
            sc = rule.fma ? (float)fma((double)p.w_progress, prog, (double)sc) : (float)((double)sc + prog);
            const float gap = madf(v, p.time_gap, p.gap_min, rule) - s_dist;
            sc = (float)madd((double)p.w_safety_dist, fmax(0.0, (double)gap), (double)sc, rule);
            st_node(term_out + (size_t)r * S + is, sc, 0.0f, 0.0f);
        }
    }
}

// ------------------------------------------------------------------------
// Best plan of a batch of scenario rollouts (new): the lexicographically smallest
// (constraint violation, cost, scenario id) -- the planner's own order (lon_planner.cu:153-170) applied to
// whole plans, with the validity measure of the Python caller (max constr over states[1:-1],
// planning/dyn_prog/poly_lat_dp_lon_planner.py:99-101) and the value of the plan (cost of the first node).
// One workgroup; the winner's trajectory is written behind a 16-byte header {constr, cost, id, t_steps}
// unless the record already holds a better one (reset == 0: running best over several chunks).
// ------------------------------------------------------------------------
__device__ __forceinline__ bool plan_better(float g, float c, int id, float bg, float bc, int bid)
{
    return g < bg || (g == bg && (c < bc || (c == bc && id < bid)));
}

__global__ void __launch_bounds__(256) k_best_plan(const TplLonState* __restrict__ trajs, int count, int id_first, int T,
                                                   unsigned char* __restrict__ record, int reset)
{
    __shared__ float s_g[256], s_c[256];
    __shared__ int s_id[256], s_take;
    const int tid = threadIdx.x;
    float bg = INFINITY, bc = INFINITY;
    int bid = 0x7fffffff;
    for (int j = tid; j < count; j += 256) {
        const TplLonState* tr = trajs + (size_t)j * T;
        float g;
        if (T > 2) { g = tr[1].constr; for (int i = 2; i < T - 1; ++i) g = fmaxf(g, tr[i].constr); }
        else { g = tr[0].constr; for (int i = 1; i < T; ++i) g = fmaxf(g, tr[i].constr); }
        const float c = tr[0].cost;
        if (plan_better(g, c, id_first + j, bg, bc, bid)) { bg = g; bc = c; bid = id_first + j; }
    }
    s_g[tid] = bg; s_c[tid] = bc; s_id[tid] = bid;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w && plan_better(s_g[tid + w], s_c[tid + w], s_id[tid + w], s_g[tid], s_c[tid], s_id[tid])) {
            s_g[tid] = s_g[tid + w]; s_c[tid] = s_c[tid + w]; s_id[tid] = s_id[tid + w];
        }
        __syncthreads();
    }
    float* rf = reinterpret_cast<float*>(record);
    int* ri = reinterpret_cast<int*>(record);
    if (tid == 0) {
        const bool have = s_id[0] != 0x7fffffff;
        s_take = have && (reset || plan_better(s_g[0], s_c[0], s_id[0], rf[0], rf[1], ri[2]));
        if (reset && !have) { rf[0] = INFINITY; rf[1] = INFINITY; ri[2] = 0x7fffffff; ri[3] = T; }
    }
    __syncthreads();
    if (!s_take) return;
    const float* src = reinterpret_cast<const float*>(trajs + (size_t)(s_id[0] - id_first) * T);
    for (int e = tid; e < T * 7; e += 256) rf[4 + e] = src[e];
    if (tid == 0) { rf[0] = s_g[0]; rf[1] = s_c[0]; ri[2] = s_id[0]; ri[3] = T; }
}

} // namespace
