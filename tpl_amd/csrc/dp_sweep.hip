// dp_sweep.hip -- the large-grid backward sweep of the DP (s,v,a) planner, third form (k_backward3v).
//
// Same data flow as k_backward2v in dp_lon.hip (two vertically adjacent rows (v, a), (v+1, a) per
// lane, six coalesced row loads per action, blended rows parked in LDS, one barrier per action), with
// the instruction stream rebuilt around what gfx950's vector ALU actually charges
// (tools/micro/valu_tp.hip, 8 waves per SIMD, nominal cycles per wave-instruction):
//
//   * every instruction takes an issue slot of ~2.5 cycles; plain fp32 add / sub / mul / fma (and
//     integer add / and) with VECTOR-register or literal operands cost nothing beyond it;
//   * everything else -- min / max, compares, selects, conversions, floor / rint, shifts, all fp64,
//     all packed-fp32 (v_pk_*), and ANY instruction with a scalar-register operand -- also occupies a
//     second unit for ~4.5 cycles, so a stream of them runs at 4.5 per instruction, and packed fp32
//     does not overlap with plain fp32 at all.
//
// k_backward2v was 1276 instructions per wave of which ~1100 were of the second kind (its weights and
// per-action constants are scalar operands, its arithmetic is packed).  Here:
//   * per-(row, action) constants that feed fp32 arithmetic are read from an LDS copy of the action
//     descriptors as broadcasts, launch constants are moved to vector registers once: the blends and
//     the coordinate / cost arithmetic are plain VGPR-operand fp32 instructions;
//   * no packed fp32 (this translation unit is built with -fno-slp-vectorize);
//   * the action-independent part of the station update, (double)(s + v dt) + a_term, is formed once
//     per cell instead of once per action;
//   * the two divisions by run-time constants are a two-float reciprocal (x*r_lo, then fma(x, r_hi, .):
//     two plain instructions) or one multiplication for a power-of-two divisor -- used only after
//     k_verify_div has compared them with IEEE division for every float of the range;
//   * the filter weight's rint(a*256)/256 is (a + 2^15) - 2^15: the sum lands where floats are 2^-8
//     apart, so the addition itself rounds to nearest-even at 8 fractional bits (exact for 0 <= a <= 1);
//   * LDS and path indices are formed as byte offsets in float (one multiplication, one conversion);
//   * the running lexicographic argmin compares (constraint, cost) as ONE unsigned 64-bit key: both
//     are sums of non-negative terms (the host checks that every weight is >= +0), and for
//     non-negative floats the integer order of the bit patterns is the float order.
// Results are bit-identical to k_backward / k_backward2v and to oracle/dp_lon_oracle.c.
#include <cstdlib>

#include "dp_lon_internal.h"

// timing experiments only (tools/ab_dp.sh build WORK -- -DV3_ABLATE=n): 1 = phase 2 removed, 2 = row loads removed
#ifndef V3_ABLATE
#define V3_ABLATE 0
#endif

namespace tplhip {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));

// pins a uniform value in a vector register (the compiler would otherwise fold the scalar register
// into the consuming instruction, which then runs on the slow path)
__device__ __forceinline__ float in_vgpr(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// x / c for a run-time constant c, in the form the host proved exact for this c (see k_verify_div)
template <int MODE>
__device__ __forceinline__ float div_const(float x, float r_hi, float r_lo)
{
    if (MODE == 4) return x * r_hi;              // c is a power of two
    return fmaf(x, r_hi, x * r_lo);              // two-float reciprocal
}

struct Node3 { float c, g, j; };

__device__ __forceinline__ Node3 lerp3(Node3 p, Node3 q, float w, float iw)      // rule R3
{
    Node3 r;
    r.c = fmaf(q.c, w, p.c * iw);
    r.g = fmaf(q.g, w, p.g * iw);
    r.j = fmaf(q.j, w, p.j * iw);
    return r;
}

// PREFETCH: the six common row loads of action k+1 are issued before the barrier of action k, so they fly under
// its phase 2 (a few more live registers).
template <int PATH_DIV, int RANGE_DIV, bool PREFETCH>
__global__ void __launch_bounds__(1024)
k_backward3v(const Node* __restrict__ next_all, Node* __restrict__ out_all, const ActDesc* __restrict__ acts,
             const float* __restrict__ path_all, const float* __restrict__ pre_vmax_all,
             const float* __restrict__ pre_sdist_all, const double* __restrict__ pre_prog, SweepArgs a)
{   // (the pointers are separate restrict-qualified parameters, not members of `a`: only then are the uniform
    // descriptor reads scalar loads -- a member pointer carries no noalias, and the barriers count as clobbers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TplDpLonParams& p = a.p;
    const int S = p.s_steps, V = p.v_steps, A = p.a_steps;
    const int nrows = V * A;
    const int is = threadIdx.x;
    const int s_pad = a.s_pad;

    // scenario of this workgroup (blockIdx.y): own tables, stage prelude and path
    const size_t scn = blockIdx.y;
    const Node* __restrict__ next = next_all + scn * a.stride_tables;
    Node* __restrict__ out = out_all + scn * a.stride_tables;
    const float* __restrict__ pre_vmax = pre_vmax_all + scn * a.stride_pre;
    const float* __restrict__ pre_sdist = pre_sdist_all + scn * a.stride_pre;
    const float* __restrict__ path = path_all + scn * a.stride_path;

    // traversal: as k_backward2v (XCD = blockIdx.x & 7 owns a v range, a-major tiles inside)
    const int xcd = blockIdx.x & 7;
    const int v_lo = (xcd * V) >> 3, v_hi = ((xcd + 1) * V) >> 3;
    const int vx = (v_hi - v_lo) >> 1, tv = max(1, a.tile_v >> 1);
    const int q = (int)(blockIdx.x >> 3);
    if (q >= vx * A) return;                                   // whole workgroup, before any barrier
    int ia, iv;
    {
        const int full = vx / tv;
        int strip = q / (A * tv), w = tv, rem = q - strip * (A * tv);
        if (strip >= full) { strip = full; w = vx - full * tv; rem = q - full * (A * tv); }
        const int tile = rem / (a.tile_a * w);
        const int r = rem - tile * (a.tile_a * w);
        ia = tile * a.tile_a + r / w;
        iv = v_lo + 2 * (strip * tv + (r - (r / w) * w));
    }
    const int row0 = ia * V + iv, row1 = row0 + 1;
    const bool active = is < S;
    const int js = min(is, S - 1);

    // LDS: float4 R[2 rows][2 buffers][s_pad + 1] | float dsc[2 rows][9][8] | float pv[path_steps + 1]
    const int row_slots = s_pad + 1;
    v4f* R = reinterpret_cast<v4f*>(smem);
    float* dsc = reinterpret_cast<float*>(smem + (size_t)4 * row_slots * sizeof(v4f));
    float* pv = dsc + 2 * N_BWD * 8;
    const ActDesc* __restrict__ d0 = acts + (size_t)row0 * N_BWD;
    const ActDesc* __restrict__ d1 = acts + (size_t)row1 * N_BWD;
    for (int i = is; i < 2 * N_BWD * 8; i += blockDim.x) {
        const int r = i / (N_BWD * 8), rem = i - r * (N_BWD * 8);
        const int k = rem >> 3, f = rem & 7;
        dsc[i] = reinterpret_cast<const float*>((r ? d1 : d0) + k)[f];
    }
    for (int i = is; i <= p.path_steps; i += blockDim.x) pv[i] = path[min(i, p.path_steps - 1) * 7 + 5];   // last sample twice

    const float s = p.s_min + (float)js * p._s_step_size;
    const float v0 = p.v_min + (float)iv * p._v_step_size, v1 = p.v_min + (float)(iv + 1) * p._v_step_size;
    const float acc = p.a_min + (float)ia * p._a_step_size;
    const float v_max = pre_vmax[js];
    const float s_dist = pre_sdist[js];
    const double prog = pre_prog[js];

    // stateCost lon_planner.cu:40-55
    float sc0, sc1, sg0, sg1;
    {
        float c0 = 0.0f;
        c0 += p.w_a * (acc * acc);
        const float x = (float)((double)c0 + prog);
        const float gap0 = (v0 * p.time_gap + p.gap_min) - s_dist;
        const float gap1 = (v1 * p.time_gap + p.gap_min) - s_dist;
        sc0 = (float)((double)x + (double)p.w_safety_dist * fmax(0.0, (double)gap0));
        sc1 = (float)((double)x + (double)p.w_safety_dist * fmax(0.0, (double)gap1));
        sg0 = 0.0f; if (v0 > v_max) sg0 += v0 - v_max;
        sg1 = 0.0f; if (v1 > v_max) sg1 += v1 - v_max;
    }
    // lonDynamics station part lon_planner.cu:244, the terms that do not depend on the action
    const double sd_base0 = (double)(s + d0->vdt) + d0->a_term;
    const double sd_base1 = (double)(s + d1->vdt) + d1->a_term;

    const __amdgpu_buffer_rsrc_t next_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<Node*>(next), 0, (int)((size_t)nrows * S * sizeof(Node)), 0x00020000);
    const unsigned voff = (unsigned)js * 12u, row_bytes = (unsigned)S * 12u;
    auto load_row = [&](int r) -> Node3 {
        const v3f x = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(next_rsrc, (int)voff, (int)((unsigned)r * row_bytes), 0));
        return Node3{x.x, x.y, x.z};
    };

    // launch constants of the fp32 arithmetic, in vector registers
    const float s_min_v = in_vgpr(p.s_min);
    const float s_scale = (float)(S - 1);
    const float s_scale_v = in_vgpr(s_scale);
    const float rr_hi = in_vgpr(a.rcp_range_hi), rr_lo = in_vgpr(a.rcp_range_lo);
    const float rp_hi = in_vgpr(a.rcp_path_hi), rp_lo = in_vgpr(a.rcp_path_lo);
    const float w_snap_v = in_vgpr(p.w_snap);
    const float path_top = (float)(p.path_steps - 1);
    const bool dup_last = is == S - 1;

    unsigned long long best0 = 0x7f8000007f800000ull, best1 = 0x7f8000007f800000ull;   // (+inf, +inf)
    int bk0 = 0, bk1 = 0;

    Node3 ld[6];
    auto issue_loads = [&](int k) {
        const ActDesc* e0 = d0 + k;
        const ActDesc* e1 = d1 + k;
        ld[0] = load_row(e0->r00); ld[1] = load_row(e0->r10); ld[2] = load_row(e0->r01); ld[3] = load_row(e0->r11);
        ld[4] = load_row(e1->r01); ld[5] = load_row(e1->r11);
    };
#if V3_ABLATE == 2       // timing experiment: no row loads at all (registers fed from lane ids), results are wrong
    for (int i = 0; i < 6; ++i) ld[i] = Node3{(float)is, (float)(is + i), 1.0f};
#endif

    __syncthreads();                     // descriptors and path limits are in LDS

    const v4f* dsc4 = reinterpret_cast<const v4f*>(dsc);
#pragma unroll
    for (int k = 0; k < N_BWD; ++k) {
        const ActDesc* e0 = d0 + k;
        const ActDesc* e1 = d1 + k;
        v4f* Rk0 = R + (k & 1) * row_slots;
        v4f* Rk1 = R + (2 + (k & 1)) * row_slots;
        // ---- phase 1: blend the next-stage rows around (v', a') for both cells (a, then v)
        {
            const v4f w0 = dsc4[(0 * N_BWD + k) * 2], w1 = dsc4[(1 * N_BWD + k) * 2];      // wa, iwa, wv, iwv
            // the upper half of row 0's window is the lower half of row 1's, blended with the same weight
            const bool shared = e0->r01 == e1->r00 && e0->r11 == e1->r10
                                && __float_as_uint(e0->wa) == __float_as_uint(e1->wa);
#if V3_ABLATE != 2
            if (!PREFETCH || k == 0) issue_loads(k);
#endif
            const Node3 p00 = ld[0], p10 = ld[1], p01 = ld[2], p11 = ld[3], q01 = ld[4], q11 = ld[5];
            const Node3 a0 = lerp3(p00, p10, w0.x, w0.y);
            const Node3 a1 = lerp3(p01, p11, w0.x, w0.y);
            Node3 b0;
            if (shared) {
                b0 = a1;
            } else {
                const Node3 q00 = load_row(e1->r00), q10 = load_row(e1->r10);
                b0 = lerp3(q00, q10, w1.x, w1.y);
            }
            const Node3 b1 = lerp3(q01, q11, w1.x, w1.y);
            const Node3 r0 = lerp3(a0, a1, w0.z, w0.w);
            const Node3 r1 = lerp3(b0, b1, w1.z, w1.w);
            *reinterpret_cast<v3f*>(Rk0 + is) = (v3f){r0.c, r0.g, r0.j};
            *reinterpret_cast<v3f*>(Rk1 + is) = (v3f){r1.c, r1.g, r1.j};
            if (dup_last) {
                *reinterpret_cast<v3f*>(Rk0 + S) = (v3f){r0.c, r0.g, r0.j};
                *reinterpret_cast<v3f*>(Rk1 + S) = (v3f){r1.c, r1.g, r1.j};
            }
#if V3_ABLATE != 2
            if (PREFETCH && k + 1 < N_BWD) {
                issue_loads(k + 1);
                __builtin_amdgcn_sched_barrier(0);            // keep the requests in front of phase 2
            }
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- phase 2: successor station, 1-D blend along s, costs, argmin -- per cell
        const v4f u0 = dsc4[(0 * N_BWD + k) * 2 + 1], u1 = dsc4[(1 * N_BWD + k) * 2 + 1];   // j, wj_cost, a_viol, tn_v
        int kv = k;
        asm volatile("" : "+v"(kv));                       // k in a vector register: the select below takes it as it is
        auto cell = [&](double sd_base, const ActDesc* e, const v4f u, const v4f* Rk, float sc, float sg,
                        unsigned long long& best, int& bk) {
            // lonDynamics station part, lon_planner.cu:244
            const double sd = sd_base + e->j_term;
            const float tn_s = fmaxf(s, (float)sd);
            // readNode coordinate along s (lon_planner.cu:35); tn_s >= s >= s_min
            const float xs = tn_s - s_min_v;
            const float qs = div_const<RANGE_DIV>(xs, rr_hi, rr_lo);
            const float cs = qs * s_scale_v + 0.5f;
            const float xb = cs - 0.5f;
            const float fl = floorf(xb);
            const float fr = xb - fl;
            const float ws = (fr + 32768.0f) - 32768.0f;          // rint(fr * 256) / 256
            const float iws = 1.0f - ws;
            const int off = (int)(fminf(fl, s_scale) * 16.0f);    // byte offset of the lower tap
            const v4f nl = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(Rk) + off);
            const v4f nh = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(Rk) + off + 16);
            asm volatile("" :: "v"(nl.w), "v"(nh.w));             // whole 16-byte reads: ds_read_b128 is twice as fast as b96
            const float nc = fmaf(nh.x, ws, nl.x * iws);
            const float ng = fmaf(nh.y, ws, nl.y * iws);
            const float nj = fmaf(nh.z, ws, nl.z * iws);
            // interpPath(tn.s).v_max, lon_planner.cu:114 (second tap = the next slot, see k_backward2)
            const float alpha = div_const<PATH_DIV>(tn_s, rp_hi, rp_lo);
            const float f0 = fminf(floorf(alpha), path_top);
            const float pa = alpha - f0;
            const float pia = 1.0f - pa;
            const int t4 = (int)(f0 * 4.0f);
            const float pv0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4);
            const float pv1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4 + 4);
            const float vmn = pv0 * pia + pv1 * pa;

            float c = sc + nc;
            float g = sg + ng;
            const float dj = nj - u.x;
            c += w_snap_v * (dj * dj);
            // lon_planner.cu:114-118 in double; max(d, 0) is the difference where the test holds, +0 otherwise
            g = (float)((double)g + fmax(e->tn_v_d - (double)vmn, 0.0));
            const float s_change = tn_s - s;
            g += fmaxf(s_change - s_dist, 0.0f);
            c += u.y;
            g += u.z;
            const unsigned long long key = ((unsigned long long)__float_as_uint(g) << 32) | __float_as_uint(c);
            const bool better = key < best;
            best = better ? key : best;
            bk = better ? kv : bk;
        };
#if V3_ABLATE != 1
        cell(sd_base0, e0, u0, Rk0, sc0, sg0, best0, bk0);
        cell(sd_base1, e1, u1, Rk1, sc1, sg1, best1, bk1);
#else
        best0 += __float_as_uint(Rk0[is].x + u0.x); best1 += __float_as_uint(Rk1[is].y + u1.x);
#endif
        asm volatile("" : "+v"(best0), "+v"(bk0), "+v"(best1), "+v"(bk1));
    }
    if (active) {
        st_node(out + (size_t)row0 * S + is, __uint_as_float((unsigned)best0), __uint_as_float((unsigned)(best0 >> 32)),
                action_jerk(p, bk0, N_BWD));
        st_node(out + (size_t)row1 * S + is, __uint_as_float((unsigned)best1), __uint_as_float((unsigned)(best1 >> 32)),
                action_jerk(p, bk1, N_BWD));
    }
}

} // namespace

size_t backward3v_lds_bytes(int s_pad, int path_steps)
{
    return (size_t)4 * (s_pad + 1) * 16 + (size_t)2 * N_BWD * 8 * sizeof(float) + sizeof(float) * ((size_t)path_steps + 1);
}

bool launch_backward3v(int path_div, int range_div, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(a.s_pad), lds_bytes, stream, a.next, a.out, a.acts, a.path, a.pre_vmax, a.pre_sdist, a.pre_prog, a);
        return true;
    };
    static const bool no_prefetch = std::getenv("TPLHIP_V3_NO_PREFETCH") != nullptr;
    if (no_prefetch) {
        if (path_div == 4 && range_div == 3) return go(k_backward3v<4, 3, false>);
        if (path_div == 3 && range_div == 3) return go(k_backward3v<3, 3, false>);
        if (path_div == 4 && range_div == 4) return go(k_backward3v<4, 4, false>);
        if (path_div == 3 && range_div == 4) return go(k_backward3v<3, 4, false>);
    }
    if (path_div == 4 && range_div == 3) return go(k_backward3v<4, 3, true>);
    if (path_div == 3 && range_div == 3) return go(k_backward3v<3, 3, true>);
    if (path_div == 4 && range_div == 4) return go(k_backward3v<4, 4, true>);
    if (path_div == 3 && range_div == 4) return go(k_backward3v<3, 4, true>);
    return false;
}

} // namespace tplhip
