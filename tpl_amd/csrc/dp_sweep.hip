// dp_sweep.hip -- the large-grid backward sweep of the DP (s,v,a) planner, third form (k_backward3v).
//
// Two vertically adjacent rows (v, a), (v+1, a) per lane, six coalesced row loads per action (action k of row v+1
// lands exactly one v' row above action k of row v, so the two 2x2 source windows share two rows), blended rows parked in LDS, with
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
// Round 1's sweep (packed fp32, per-action constants as scalar operands; removed in round 3) was 1276 instructions
// per wave of which ~1100 were of the second kind.  Here:
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
// Results are bit-identical to k_backward and to oracle/dp_lon_oracle.c.
#include <cstdlib>

#include "dp_lon_internal.h"

// Measured choices of the two kernels below (same-box A/Bs and the timing experiments that priced them -- phase 2
// removed, row loads removed, barriers removed, fewer loads per action, ... -- in profiles/r03_dp_sweep_ablation.txt;
// the preprocessor switches those builds used are gone from this file, `git log -S V3_ABLATE` finds them):
//  * wave priority 1 while a wave requests and blends its rows (phase 1), 0 in phase 2: the loads go out and the block
//    reaches its barrier first (0.9 %);
//  * the scalar descriptor words of action k+1 (row ids, j_term, tn_v) are requested at the start of phase 2 of action
//    k: neither the row loads nor the first instruction of a phase 2 wait for a scalar-memory round trip;
//  * the three a0 rows of action k+1 are requested from the first cell of phase 2 of action k, those of action 0 at the
//    top of the kernel ahead of the descriptor / path copies to LDS (no request ahead: +6 us per 256x128x64 sweep; more
//    rows or two actions ahead cost the eighth wavefront per SIMD and lose);
//  * actions whose blend along a is trivial (ACT_SINGLE_A, dp_lon_internal.h) request and blend their a0 rows only:
//    6 -> 3 row loads for a third to a half of all actions (40.85 -> 39.79 us; k_backward3s: 4 -> 2, in the instance
//    that does not request its first round early).  A conditional request that CROSSES a phase costs this compiler 8-14
//    registers = a wavefront per SIMD: the a1 rows only exist inside the arm of phase 1 that blends them.
// Round 4 (profiles/r04_dp_sweep_ablation.txt, blocks 12 and 13; both switches are gone again: `git log -S TPL_SWEEP_VGUARD`
// finds the first, profiles/r04_a1_lds_staging.patch is the second):
//  * a wave-uniform guard around the fp64 speed-limit term -- `if (ballot(tn_v > vmn)) g = (float)((double)g + fmax(tn_v_d
//    - vmn, 0))`, exact because adding +0.0 in double and rounding back is the identity -- LOSES: 39.47 -> 39.91 us per
//    256x128x64 sweep, 10.75 -> 10.83 on 128x64x32.  With v up to 36 m/s against limits of 9-14 m/s most rows exceed the
//    limit in every lane: compare and branch are paid everywhere, the six fp64 instructions saved on a third of the rows;
//  * the a1 rows of the NEXT action requested a phase ahead straight into LDS (gfx950's `buffer_load_dwordx3 ... lds`,
//    no vector register): lane i's 12 bytes land at M0 + 16 i (tools/micro/lds_direct_probe.hip), i.e. 3 KB per wavefront
//    for the three rows, 21.9 KB per workgroup = seven workgroups per compute unit instead of eight: 37.96 -> 41.58 us
//    (256x128x64), 319.6 -> 350.3 (512x256x128), bit-identical.  The sweep is at its plateau for this decomposition.


#ifndef TPL_CHAIN_SLEEP
#define TPL_CHAIN_SLEEP 8      /* x 64 cycles between two polls of a chained sweep's wait (A/B'd: profiles/r06_dp_chained_sweep.txt) */
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

// LDS byte addresses of the per-cell look-ups (blended-row slot, path sample): index x scale + BASE formed in float by ONE fma
// and converted -- a multiplication, a conversion and an integer add otherwise, i.e. two VALU instructions less per cell and
// action (round 5, same box: 40.75 -> 39.55 us per 256x128x64 sweep alone, 35.4 -> 34.3 with two plans in flight;
// profiles/r05_dp_addr_fma_ab.txt).  The operands are integers below 2^17, so the fma is exact; the base is the low half of
// the generic address (= the LDS offset), pinned in a vector register.  k_backward3v only: k_backward3s (a lone workgroup's chain,
// not VALU issue) gains nothing from it, k_backward3o has no register to spare (spills / a wavefront per SIMD less).
typedef const __attribute__((address_space(3))) v4f* LdsV4;
typedef const __attribute__((address_space(3))) float* LdsF;
__device__ __forceinline__ float lds_base_f(const void* q) { return in_vgpr((float)(unsigned)(unsigned long long)q); }
__device__ __forceinline__ unsigned lds_addr(float idx, float scale, float base_f) { return (unsigned)fmaf(idx, scale, base_f); }

// x / c for a run-time constant c, in the cheapest form the host proved exact for this c (see k_verify_div):
//   4  c is a power of two: one multiplication by 1/c
//   3  two-float reciprocal r_hi + r_lo: fma(x, r_hi, x * r_lo)                             (two plain instructions)
//   5  one Newton step on the quotient: q = x r; e = fma(-q, c, x); q' = fma(e, r, q)      (three; aux carries c)
//   0  IEEE division (aux carries c): every divisor, ~10 instructions -- what keeps a grid whose divisors defeat all
//      three proofs on the plain-fp32 sweeps instead of sending it to the general kernel
template <int MODE>
__device__ __forceinline__ float div_const(float x, float r_hi, float aux)
{
    if (MODE == 4) return x * r_hi;
    if (MODE == 3) return fmaf(x, r_hi, x * aux);
    if (MODE == 5) { const float q = x * r_hi; const float e = fmaf(-q, aux, x); return fmaf(e, r_hi, q); }
    return x / aux;
}

// The path-step division: the reciprocal forms are proven for the dividends 0 and [2^-96, max] (k_verify_div); a
// wavefront in which any lane holds a positive dividend below that -- which takes a successor station that cancels to
// within 1e-29 m of the road's origin -- divides the IEEE way.  One integer add, one compare and a branch never taken.
template <int MODE>
__device__ __forceinline__ float div_path(float x, float r_hi, float aux, float c)
{
    if (MODE == 3 || MODE == 5) {
        const bool tiny = (__float_as_uint(x) - 1u) < (0x0F800000u - 1u);        // 0 < x < 2^-96 (x >= 0 here)
        if (__builtin_amdgcn_ballot_w64(tiny) != 0ull) {
            asm volatile("; tiny dividend: IEEE division");      // (a side effect: keeps the division BEHIND the branch -- without it hipcc
            return x / c;                                          //  computes both forms for every cell and selects: 39.3 -> 43.1 us)
        }
    }
    return div_const<MODE>(x, r_hi, aux);
}

struct Node3 { float c, g, j; };

// rule R3 (dp_lon_math.h, lerp1): 0 = fma(q, w, p (1 - w)), 1 = p (1 - w) + q w with every operation rounded, 2 = p + w (q - p)
// (the template argument LERP of this file also carries rule R1 in bit 2 -- LERP = blend form + 4 x contraction -- so that
// the kernels keep four template parameters; kFma<LERP> reads it)
template <int LERP> constexpr bool kFma = (LERP & 4) != 0;

template <int LERP>
__device__ __forceinline__ float blend1(float p, float q, float w, float iw)
{
    if ((LERP & 3) == 1) { const float a = p * iw, b = q * w; return a + b; }
    if ((LERP & 3) == 2) { const float d = q - p; const float e = w * d; return p + e; }
    return fmaf(q, w, p * iw);
}

// a*b + c under rule R1: unfused (two roundings; the file is built -ffp-contract=off) or contracted
template <int LERP> __device__ __forceinline__ float mad1(float a, float b, float c) { return kFma<LERP> ? fmaf(a, b, c) : a * b + c; }
template <int LERP> __device__ __forceinline__ double mad1(double a, double b, double c) { return kFma<LERP> ? fma(a, b, c) : a * b + c; }

template <int LERP>
__device__ __forceinline__ Node3 lerp3(Node3 p, Node3 q, float w, float iw)
{
    Node3 r;
    r.c = blend1<LERP>(p.c, q.c, w, iw);
    r.g = blend1<LERP>(p.g, q.g, w, iw);
    r.j = blend1<LERP>(p.j, q.j, w, iw);
    return r;
}

// rule R2 (quant_weight): the filter weight of the blend along s from the fraction fr in [0, 1)
//   1  rint(fr * 256) / 256 as (fr + 2^15) - 2^15: the sum lands where floats are 2^-8 apart, so the addition itself
//      rounds to nearest-even at 8 fractional bits;   2  floor(fr * 256) / 256;   0  not quantised
template <int QUANT>
__device__ __forceinline__ float s_weight(float fr)
{
    if (QUANT == 1) return (fr + 32768.0f) - 32768.0f;
    if (QUANT == 2) return floorf(fr * 256.0f) * 0.00390625f;
    return fr;
}

// ------------------------------------------------------------------------
// NV = 2 vertically adjacent rows (v, a), (v+1, a) per lane.  For action k the two rows blend the same a' pair one
// v' row apart, so "upper a-blend of row 0" IS "lower a-blend of row 1" (checked per action on the descriptors:
// rows and weight equal): 2 + 2 NV row loads per action for NV cells and NV + 1 a-blends.  The code is written for
// general NV; NV = 4 (2.5 loads per cell instead of 3) was measured and lost, 54 us against 42: its 100 registers
// leave four waves per SIMD.  ONE LDS row buffer per row and two barriers per action: the kernel fits 64 registers,
// i.e. eight waves per SIMD, which beat the two-buffer / one-barrier / loads-in-flight-across-the-barrier form
// (68-70 registers, seven waves) by 2 % -- 42.1 against 42.9 us per 256x128x64 sweep on the same box.
// ------------------------------------------------------------------------
// CHAIN: the chained sweep (SweepChain, dp_lon_internal.h) -- a template parameter, not a run-time branch: with the branch in it
// the plain launch ran 4-5 % slower on the same box (38.0-38.8 -> 39.9-40.9 us, profiles/r06_dp_chained_sweep.txt); built for the
// default filter rule under both settings of R1.
template <int PATH_DIV, int RANGE_DIV, int QUANT, int LERP, bool CHAIN = false>
__global__ void __launch_bounds__(1024)
k_backward3v(const Node* __restrict__ next_all, Node* __restrict__ out_all, const ActDesc* __restrict__ acts,
              const float* __restrict__ path_all, const float* __restrict__ pre_vmax_all,
              const float* __restrict__ pre_sdist_all, const double* __restrict__ pre_prog, SweepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TplDpLonParams& p = a.p;
    const int S = p.s_steps, V = p.v_steps, A = p.a_steps;
    const int nrows = V * A;
    const int is = threadIdx.x;
    const int s_pad = a.s_pad;
    // blockIdx.y: the scenario of a batch, or -- chained sweep (SweepChain) -- how many stages behind the launch's first one
    constexpr bool chained = CHAIN;
    const size_t scn = chained ? 0 : blockIdx.y;
    const int cstage = chained ? (int)blockIdx.y : 0;
    const size_t back = chained ? (size_t)cstage * a.chain.table_cells : 0;
    const Node* __restrict__ next = next_all + scn * a.stride_tables - back;
    Node* __restrict__ out = out_all + scn * a.stride_tables - back;
    const float* __restrict__ pre_vmax = pre_vmax_all + scn * a.stride_pre - (size_t)cstage * S;
    const float* __restrict__ pre_sdist = pre_sdist_all + scn * a.stride_pre - (size_t)cstage * S;
    const float* __restrict__ path = path_all + scn * a.stride_path;

    constexpr int NV = 2;
    // traversal: v3_map_block (dp_lon_internal.h) -- read from the table k_fill_v3_map wrote behind the action table, or computed.
    // The table's entry also carries the three next-stage rows action 0 asks for first (V3Map): ONE scalar round trip between
    // the kernel arguments and the workgroup's first row requests instead of two (traversal word, then the descriptor's rows).
    int ia, iv;
    int first_rows[3];
    const bool mapped = a.tile_map != nullptr;
    if (mapped) {
        // (one s_load_dwordx4: as four members the compiler loads `where`, branches, and only then asks for the rows)
        typedef int v4i __attribute__((ext_vector_type(4)));
        v4i m = *reinterpret_cast<const v4i*>(a.tile_map + blockIdx.x);
        asm volatile("" : "+s"(m));
        if ((unsigned)m.x == V3_MAP_NONE) return;
        ia = (int)((unsigned)m.x >> 16); iv = (int)((unsigned)m.x & 0xffffu);
        first_rows[0] = m.y; first_rows[1] = m.z; first_rows[2] = m.w;
    } else if (!v3_map_block((int)blockIdx.x, V, A, a.tile_v, a.tile_a, a.skew_q16, ia, iv)) {
        return;
    }
    const int row0 = ia * V + iv;
    const int rs = (iv + 1 < V) ? N_BWD : 0;            // descriptor stride to the lane's second row (0: the row twice)
    const bool active = is < S;
    const int js = min(is, S - 1);

    // LDS: float4 R[NV rows][s_pad + 1] | float dsc[NV rows][9][8] | float pv[path_steps + 1]
    const int row_slots = s_pad + 1;
    v4f* R = reinterpret_cast<v4f*>(smem);
    float* dsc = reinterpret_cast<float*>(smem + (size_t)NV * row_slots * sizeof(v4f));
    float* pv = dsc + NV * N_BWD * 8;
    const ActDesc* __restrict__ d0 = acts + (size_t)row0 * N_BWD;
    const __amdgpu_buffer_rsrc_t next_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<Node*>(next), 0, (int)((size_t)nrows * S * sizeof(Node)), 0x00020000);
    const unsigned voff = (unsigned)js * 12u, row_bytes = (unsigned)S * 12u;
    auto load_row = [&](int r) -> Node3 {
        const v3f x = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(next_rsrc, (int)voff, (int)((unsigned)r * row_bytes), 0));
        return Node3{x.x, x.y, x.z};
    };
    Node3 lo[2], up[NV][2];
    // scalar descriptor words of ONE action: the rows (lower pair of row 0, upper pair of every row) and the two doubles
    struct ActScalars { int r00, r10, r01[NV], r11[NV], single, shared; double j_term[NV], tn_v_d[NV]; };
    auto fetch_scalars = [&](int k) {
        ActScalars x;
        x.r00 = d0[k].r00; x.r10 = d0[k].r10;
        const int fl = d0[k].flags;
        x.single = fl & ACT_SINGLE_A;               // (a property of (a, k): the same for both rows of the pair)
        x.shared = fl & ACT_SHARED_UP;              // row 1's lower a-blend IS row 0's upper one (k_build_actions compared the descriptors)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const ActDesc& e = d0[i * rs + k];
            x.r01[i] = e.r01; x.r11[i] = e.r11; x.j_term[i] = e.j_term; x.tn_v_d[i] = e.tn_v_d;
        }
        return x;
    };
    // row requests by mask (bit 0 lo[0], 1 lo[1], 2 + 2i up[i][0], 3 + 2i up[i][1]); A0_ROWS / A1_ROWS: the rows of the lower / upper a
    constexpr unsigned A0_ROWS = 0x15u, A1_ROWS = 0x2au;
    static_assert(NV == 2, "row masks are written for two rows per lane");
    auto issue = [&](const ActScalars& x, unsigned m) {
        if (m & 1u) lo[0] = load_row(x.r00);
        if (m & 2u) lo[1] = load_row(x.r10);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if ((m >> (2 + 2 * i)) & 1u) up[i][0] = load_row(x.r01[i]);
            if ((m >> (3 + 2 * i)) & 1u) up[i][1] = load_row(x.r11[i]);
        }
    };
    // chained sweep: the next stage's table is being written by this very launch -- this workgroup's rows are requested behind
    // the wait at the first barrier (chain_wait below); what it waits for is asked for here
    const bool wait_first = chained && cstage > 0;
    int dep = -1;
    if (wait_first && is < 64) dep = a.chain.deps[(size_t)blockIdx.x * CHAIN_DEPS + is];
    // the rows of action 0 are requested before anything else: their latency passes under the prologue
    static_assert(A0_ROWS == 0x15u, "V3Map carries lo[0], up[0][0], up[1][0]");
    if (mapped && !wait_first) {
        lo[0] = load_row(first_rows[0]); up[0][0] = load_row(first_rows[1]); up[1][0] = load_row(first_rows[2]);
        __builtin_amdgcn_sched_barrier(0);           // (in front of the descriptor's scalar words, which nothing here waits for)
    }
    ActScalars cur = fetch_scalars(0);
    if (!mapped) issue(cur, A0_ROWS);
    // ONE memory round trip for the prologue (see k_backward3s): requests first, LDS stores after
    const int bd = (int)blockDim.x;
    auto dsc_word = [&](int i) { return reinterpret_cast<const float*>(d0 + (i / (N_BWD * 8)) * rs + ((i >> 3) % N_BWD))[i & 7]; };
    float dsc_v = 0.0f, pv_v = 0.0f;
    if (is < NV * N_BWD * 8) dsc_v = dsc_word(is);
    if (is <= p.path_steps) pv_v = path[min(is, p.path_steps - 1) * 7 + 5];
    const float s = mad1<LERP>((float)js, p._s_step_size, p.s_min);
    const float acc = mad1<LERP>((float)ia, p._a_step_size, p.a_min);
    const float v_max = pre_vmax[js];
    const float s_dist = pre_sdist[js];
    const double prog = pre_prog[js];                // (contracted arithmetic: the multiplicand |1000 - s|, see k_stage_prelude)
    if (is < NV * N_BWD * 8) dsc[is] = dsc_v;
    if (is <= p.path_steps) pv[is] = pv_v;
    for (int i = is + bd; i < NV * N_BWD * 8; i += bd) dsc[i] = dsc_word(i);                   // rows of 64 / 128 lanes
    for (int i = is + bd; i <= p.path_steps; i += bd) pv[i] = path[min(i, p.path_steps - 1) * 7 + 5];
    float sc[NV], sg[NV];
    double sd_base[NV];
    {
        float c0 = 0.0f;
        c0 += p.w_a * (acc * acc);
        const float x = kFma<LERP> ? (float)fma((double)p.w_progress, prog, (double)c0) : (float)((double)c0 + prog);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float v = mad1<LERP>((float)(iv + (rs ? i : 0)), p._v_step_size, p.v_min);
            const float gap = mad1<LERP>(v, p.time_gap, p.gap_min) - s_dist;
            sc[i] = (float)mad1<LERP>((double)p.w_safety_dist, fmax(0.0, (double)gap), (double)x);
            sg[i] = 0.0f; if (v > v_max) sg[i] += v - v_max;
            // station update, lon_planner.cu:244.  Contracted, the descriptor carries the factors in front of the fused dt
            // (k_build_actions): vdt = v, a_term = (a / 2) dt, j_term = ((j / 6) dt) dt
            sd_base[i] = kFma<LERP> ? fma(d0[i * rs].a_term, (double)p.dt, (double)fmaf(d0[i * rs].vdt, p.dt, s))
                                    : (double)(s + d0[i * rs].vdt) + d0[i * rs].a_term;
        }
    }
    static_assert(NV == 2, "ACT_SHARED_UP describes the pair (v, v + 1)");
    // (which actions let row 1 reuse row 0's upper a-blend is a bit of the descriptor since round 6 -- ACT_SHARED_UP, tested by
    // k_build_actions once per reinit.  Until then every workgroup compared the two rows' descriptors itself: up to 27 DEPENDENT
    // scalar loads in its prologue; requesting them all at once instead -- 100 scalar registers -- lost, 38.5 -> 39.5 us per
    // 256x128x64 sweep and 315 -> 333 on 512x256x128; profiles/r06_dp_contraction_ab.txt)
    const float s_min_v = in_vgpr(p.s_min);
    const float s_scale = (float)(S - 1);
    const float s_scale_v = in_vgpr(s_scale);
    const float rr_hi = in_vgpr(a.rcp_range_hi), rr_lo = in_vgpr(a.rcp_range_lo);
    const float rp_hi = in_vgpr(a.rcp_path_hi), rp_lo = in_vgpr(a.rcp_path_lo);
    const float w_snap_v = in_vgpr(p.w_snap);
    // (three registers: the instances whose path division needs a reciprocal pair sit at 72-75 registers; those at 72 -- the
    // station range divided by one multiplication or the IEEE way -- hold seven wavefronts per SIMD, would lose one, and keep the
    // integer form; the others hold six either way)
    constexpr bool ADDR_FMA = PATH_DIV == 4 || PATH_DIV == 0 || RANGE_DIV == 3 || RANGE_DIV == 5;
    float rk_base_f[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) rk_base_f[i] = ADDR_FMA ? lds_base_f(R + i * row_slots) : 0.0f;
    const float pv_base_f = ADDR_FMA ? lds_base_f(pv) : 0.0f;
    const float path_top = (float)(p.path_steps - 1);
    const double dt_d = (double)p.dt;
    const bool dup_last = is == S - 1;
    unsigned long long best[NV];
    int bk[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { best[i] = 0x7f8000007f800000ull; bk[i] = 0; }
    // The per-cell state terms are finished HERE, in front of the barrier (an empty `asm` on their registers).  Left alone the
    // compiler sinks the whole block -- 35 instructions, 50 with the fp64 fma chains of contracted arithmetic -- into phase 1 of
    // action 0, between a wave's row stores and the barrier its workgroup waits at: every workgroup's first barrier is late by
    // that chain.  Round 6, same box, alone on the chip: R1 = 1 39.5-40.0 -> 38.4-38.8 us per 256x128x64 sweep, R1 = 0
    // 38.9-39.5 -> 38.5-39.0 (profiles/r06_dp_contraction_ab.txt) -- what made the contracted instance, which has FEWER
    // instructions, the slower of the two.
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(sc[i]), "+v"(sg[i]), "+v"(sd_base[i]));
    if (wait_first) {
        // chain_wait: the first wavefront polls the flags of the row pairs the nine actions read (agent scope: past this XCD's
        // L2) until all carry this launch's epoch; it gives up after `budget` ticks of the 100 MHz counter, or as soon as any
        // workgroup has, and says so in *err -- the host then discards the plan (dp_lon.hip: chain_check)
        if (is < 64) {
            const unsigned* fl = a.chain.flags + (size_t)(a.chain.first - cstage + 1) * a.chain.pairs;
            const long long t0 = (long long)wall_clock64();
            for (;;) {
                const bool ok = dep < 0 || __hip_atomic_load(fl + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.chain.wait_epoch;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                const long long waited = (long long)wall_clock64() - t0;
                if (waited > 1000 && (waited > a.chain.budget || __hip_atomic_load(a.chain.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    if (is == 0) {
                        __hip_atomic_store(a.chain.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(a.chain.err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(TPL_CHAIN_SLEEP);
            }
#ifdef TPL_CHAIN_STATS
            if (is == 0) {
                const long long waited = (long long)wall_clock64() - t0;
                atomicAdd(reinterpret_cast<unsigned long long*>(a.chain.err + 2), (unsigned long long)waited);
                if (waited > 300) atomicAdd(a.chain.err + 4, 1u);
                atomicAdd(a.chain.err + 5, 1u);
            }
#endif
        }
    }
    __syncthreads();                     // descriptors and path limits are in LDS (chained: and the rows this workgroup reads are complete)
    if (wait_first) {
        lo[0] = load_row(first_rows[0]); up[0][0] = load_row(first_rows[1]); up[1][0] = load_row(first_rows[2]);
    }
    // (the descriptor copy's LDS address stays in ONE vector register where there is one to spare: otherwise it is moved there
    // from a scalar one per action)
    unsigned dsc_addr = (unsigned)(unsigned long long)dsc;
    if constexpr (ADDR_FMA) asm volatile("" : "+v"(dsc_addr));
    const LdsV4 dsc4 = (LdsV4)(unsigned long long)dsc_addr;
#pragma unroll
    for (int k = 0; k < N_BWD; ++k) {
        // ---- phase 1: blend the next-stage rows around (v', a') for the NV cells (a, then v)
        {
            __builtin_amdgcn_s_setprio(1);
            // (the a0 rows were requested during phase 2 of action k-1, those of action 0 in the prologue)
            // The blend along a is trivial for a third to a half of all (a, action) pairs -- the successor acceleration is
            // clamped to the first / last a-row (both taps are the same row) or lands on an a-row (weight exactly 0, see
            // ACT_SINGLE_A): such an action requests its three a0 rows only and blends them with themselves.  The a0 rows
            // of every action are requested one phase ahead (they are needed either way); the a1 rows only exist inside
            // the arm that loads them, so no register of theirs crosses an action.
            v4f w[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) w[i] = dsc4[(i * N_BWD + k) * 2];
            // (barrier and stores inside the arms: results merged behind the branch cost a move per register)
            const auto park = [&](const Node3 (&r)[NV]) {
                if (k > 0) asm volatile("s_barrier" ::: "memory");      // everyone is done reading the row buffers of action k-1
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    v4f* Rk = R + i * row_slots;
                    *reinterpret_cast<v3f*>(Rk + is) = (v3f){r[i].c, r[i].g, r[i].j};
                    if (dup_last) *reinterpret_cast<v3f*>(Rk + S) = (v3f){r[i].c, r[i].g, r[i].j};
                }
            };
            if (!cur.single) {
                Node3 r[NV];
                issue(cur, A1_ROWS);
                Node3 bl = lerp3<LERP>(lo[0], lo[1], w[0].x, w[0].y);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const Node3 bu = lerp3<LERP>(up[i][0], up[i][1], w[i].x, w[i].y);
                    r[i] = lerp3<LERP>(bl, bu, w[i].z, w[i].w);
                    // (the cell's row is finished HERE, ahead of the shared / not-shared branch: left to the compiler it sinks behind
                    // the branch, keeps both blends alive across it and pays three moves per action and four registers for that)
                    if (i + 1 < NV) asm volatile("" : "+v"(r[i].c), "+v"(r[i].g), "+v"(r[i].j));
                    if (i + 1 < NV) {
                        if (cur.shared) {
                            bl = bu;
                        } else {
                            const ActDesc& y = d0[(i + 1) * rs + k];
                            const Node3 q00 = load_row(y.r00), q10 = load_row(y.r10);
                            bl = lerp3<LERP>(q00, q10, w[i + 1].x, w[i + 1].y);
                        }
                    }
                }
                park(r);
            } else if (__builtin_amdgcn_readfirstlane((int)__float_as_uint(w[0].x)) == 0) {
                // a weight of exactly 0 along a (the jerk grid lands on the acceleration grid: 106 of the 576 (a, action) pairs of
                // 256x128x64): the blend of a row with itself IS the row -- fma(p, 0, p * 1) = p for every finite p in every blend form,
                // see ACT_SINGLE_A -- so the three blends along a are not computed (a wave-uniform test; same box, two plans in flight:
                // 34.4 -> 33.9 us per sweep, nothing for a lone launch; profiles/r05_dp_addr_fma_ab.txt, call 6)
                Node3 r[NV];
                Node3 bl = lo[0];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const Node3 bu = up[i][0];
                    r[i] = lerp3<LERP>(bl, bu, w[i].z, w[i].w);
                    if (i + 1 < NV) asm volatile("" : "+v"(r[i].c), "+v"(r[i].g), "+v"(r[i].j));
                    if (i + 1 < NV) {
                        if (cur.shared) bl = bu;
                        else bl = load_row(d0[(i + 1) * rs + k].r00);
                    }
                }
                park(r);
            } else {
                Node3 r[NV];
                Node3 bl = lerp3<LERP>(lo[0], lo[0], w[0].x, w[0].y);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const Node3 bu = lerp3<LERP>(up[i][0], up[i][0], w[i].x, w[i].y);
                    r[i] = lerp3<LERP>(bl, bu, w[i].z, w[i].w);
                    // (the cell's row is finished HERE, ahead of the shared / not-shared branch: left to the compiler it sinks behind
                    // the branch, keeps both blends alive across it and pays three moves per action and four registers for that)
                    if (i + 1 < NV) asm volatile("" : "+v"(r[i].c), "+v"(r[i].g), "+v"(r[i].j));
                    if (i + 1 < NV) {
                        if (cur.shared) {
                            bl = bu;
                        } else {
                            const Node3 q00 = load_row(d0[(i + 1) * rs + k].r00);
                            bl = lerp3<LERP>(q00, q00, w[i + 1].x, w[i + 1].y);
                        }
                    }
                }
                park(r);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_s_setprio(0);
        // ---- phase 2: successor station, 1-D blend along s, costs, argmin -- per cell
        int kv = k;
        asm volatile("" : "+v"(kv));
        const ActScalars now = cur;
        if (k + 1 < N_BWD) {
            cur = fetch_scalars(k + 1);
            __builtin_amdgcn_sched_barrier(0);                 // the requests go out before this phase's arithmetic
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const v4f u = dsc4[(i * N_BWD + k) * 2 + 1];
            // lonDynamics station part, lon_planner.cu:244
            const double sd = kFma<LERP> ? fma(now.j_term[i], dt_d, sd_base[i]) : sd_base[i] + now.j_term[i];
            const float tn_s = fmaxf(s, (float)sd);
            const float xs = tn_s - s_min_v;
            const float qs = div_const<RANGE_DIV>(xs, rr_hi, rr_lo);
            const float cs = qs * s_scale_v + 0.5f;
            const float xb = cs - 0.5f;
            const float fl = floorf(xb);
            const float fr = xb - fl;
            const float ws = s_weight<QUANT>(fr);
            const float iws = 1.0f - ws;
            v4f nl, nh;
            if constexpr (ADDR_FMA) {
                const unsigned off = lds_addr(fminf(fl, s_scale), 16.0f, rk_base_f[i]);
                nl = *(LdsV4)(unsigned long long)off;
                nh = *(LdsV4)(unsigned long long)(off + 16u);
            } else {
                const int off = (int)(fminf(fl, s_scale) * 16.0f);
                const char* Rk = reinterpret_cast<const char*>(R + i * row_slots);
                nl = *reinterpret_cast<const v4f*>(Rk + off);
                nh = *reinterpret_cast<const v4f*>(Rk + off + 16);
            }
            asm volatile("" :: "v"(nl.w), "v"(nh.w));
            const float nc = blend1<LERP>(nl.x, nh.x, ws, iws);
            const float ng = blend1<LERP>(nl.y, nh.y, ws, iws);
            const float nj = blend1<LERP>(nl.z, nh.z, ws, iws);
            // The first rows of the NEXT action go out here, in the first cell of this one: their registers are free (the
            // blends of phase 1 consumed them), the scalar row ids requested at the top of the phase have arrived with
            // the LDS reads just consumed (one counter), and their latency passes under the rest of the phase.
            if (i == 0 && k + 1 < N_BWD) {
                __builtin_amdgcn_sched_barrier(0);
                issue(cur, A0_ROWS);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float alpha = div_path<PATH_DIV>(tn_s, rp_hi, rp_lo, p.path_step_size);
            const float f0 = fminf(floorf(alpha), path_top);
            const float pa = alpha - f0;
            const float pia = 1.0f - pa;
            float pv0, pv1;
            if constexpr (ADDR_FMA) {
                const unsigned t4 = lds_addr(f0, 4.0f, pv_base_f);
                pv0 = *(LdsF)(unsigned long long)t4;
                pv1 = *(LdsF)(unsigned long long)(t4 + 4u);
            } else {
                const int t4 = (int)(f0 * 4.0f);
                pv0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4);
                pv1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4 + 4);
            }
            const float vmn = mad1<LERP>(pv0, pia, pv1 * pa);
            float c = sc[i] + nc;
            float g = sg[i] + ng;
            const float dj = nj - u.x;
            c = mad1<LERP>(w_snap_v, dj * dj, c);
            g = (float)((double)g + fmax(now.tn_v_d[i] - (double)vmn, 0.0));
            const float s_change = tn_s - s;
            g += fmaxf(s_change - s_dist, 0.0f);
            c += u.y;
            g += u.z;
            const unsigned long long key = ((unsigned long long)__float_as_uint(g) << 32) | __float_as_uint(c);
            const bool better = key < best[i];
            best[i] = better ? key : best[i];
            bk[i] = better ? kv : bk[i];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(best[i]), "+v"(bk[i]));
    }
    if (!chained) {
        if (active) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (i == 0 || rs) st_node(out + (size_t)(row0 + i) * S + is, __uint_as_float((unsigned)best[i]), __uint_as_float((unsigned)(best[i] >> 32)),
                        action_jerk(p, bk[i], N_BWD, FilterRule{QUANT, LERP & 3, kFma<LERP> ? 1 : 0}));
        }
    } else {
        // chained: the nodes go THROUGH this XCD's L2 (agent scope), and once every wave's stores have completed the pair is announced
        if (active) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (i == 0 || rs) {
                    const v3f n = {__uint_as_float((unsigned)best[i]), __uint_as_float((unsigned)(best[i] >> 32)),
                                   action_jerk(p, bk[i], N_BWD, FilterRule{QUANT, LERP & 3, kFma<LERP> ? 1 : 0})};
                    Node* q = out + (size_t)(row0 + i) * S + is;
                    asm volatile("global_store_dwordx3 %0, %1, off sc1" :: "v"(q), "v"(n) : "memory");
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (is == 0)
            __hip_atomic_store(a.chain.flags + (size_t)(a.chain.first - cstage) * a.chain.pairs + (size_t)ia * ((V + 1) / 2) + (iv >> 1),
                               a.chain.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------
// k_backward3o: k_backward3v for rule R3 under ANOTHER ORDER (tplhip_dp_lon_set_blend_order; FilterRule::order):
//   ORDER 1  three sequential blends s -> v -> a (the texture's x first),
//   ORDER 2  the symmetric sum of w_c T_c over the eight corners, fp32 accumulator in the documented term order,
//   ORDER 3  the same sum with exact products in a double accumulator, one rounding.
// None of them lets a row be blended ahead of the station -- the station weight is the per-cell one and comes first (order 1)
// or multiplies every corner (the sums) -- so phase 1 parks the six RAW rows of the pair and every cell-action takes its
// eight taps from LDS.  Same geometry, traversal, prologue, division forms, cost arithmetic and integer-key argmin as
// k_backward3v; what changes is the look-up:
//   * LDS row block: one slot of 19 floats per station = 6 rows x (cost, constr, jerk) + one float of padding (an odd
//     stride in dwords: conflict-free for the shifted reads) -- rows B0 (a0,v0) B1 (a1,v0) B2 (a0,v1) B3 (a1,v1) B4 (a0,v2)
//     B5 (a1,v2); cell 0 blends B0..B3, cell 1 B2..B5 (the pair shares its middle v' row, see k_backward3v).  Both taps of a
//     channel of a row are ONE ds_read2_b32 (offsets c and 19 + c): 12 LDS reads per cell-action against two.  (A channel-major
//     slot of 28 floats -- the four rows of a cell as one 16-byte read per channel and tap, six reads instead of twelve for the
//     same bytes -- measured 78.9 / 79.3 / 84.6 us against 63.5 / 65.2 / 73.9: the wide reads of neighbouring lanes collide.)
//   * Where the descriptors of the two rows do NOT share (rounding of the successor velocity near a grid row, the clamped
//     edges), cell 1's own lower rows are parked over B2, B3 between the two cells (a workgroup is one row pair, so the
//     case is uniform: two more barriers for that action).
//   * ACT_SINGLE_A (the blend along a is trivial: twin a-rows, or a weight of exactly 0): the three a0 rows only are requested
//     and parked, and a cell reads four taps per channel pair (fetch8_single) -- exact, see there; 63.2 -> 59.9 us (order 1).
//   * Six wavefronts per SIMD for the sequential order (80 registers without a spill: 59.9 -> 55.2 us); the sums need 83-90
//     registers and stay at five.
// Built for the weight rules rint and floor (an un-quantised weight makes the corner products inexact and takes the general
// sweep), both contraction settings, and for order 1 all three blend forms.  Price: DESIGN.md 0.
// ------------------------------------------------------------------------
constexpr int SLOT_DW = 19;

template <int ORDER, int LERP>
__device__ __forceinline__ float fetch8(const float (&lo)[4], const float (&hi)[4], float ws, float iws, const v4f& w)
{
    // lo / hi: the taps at stations i, i + 1 of rows B0 (a0,v0) B1 (a1,v0) B2 (a0,v1) B3 (a1,v1); w = (wa, 1 - wa, wv, 1 - wv)
    if (ORDER == 1) {
        const float x00 = blend1<LERP>(lo[0], hi[0], ws, iws), x10 = blend1<LERP>(lo[1], hi[1], ws, iws);
        const float x01 = blend1<LERP>(lo[2], hi[2], ws, iws), x11 = blend1<LERP>(lo[3], hi[3], ws, iws);
        const float y0 = blend1<LERP>(x00, x01, w.z, w.w), y1 = blend1<LERP>(x10, x11, w.z, w.w);
        return blend1<LERP>(y0, y1, w.x, w.y);
    }
    // corner c = s + 2 v + 4 a, weight (w_s * w_v) * w_a
    const float sv00 = iws * w.w, sv10 = ws * w.w, sv01 = iws * w.z, sv11 = ws * w.z;
    const float wc[8] = {sv00 * w.y, sv10 * w.y, sv01 * w.y, sv11 * w.y, sv00 * w.x, sv10 * w.x, sv01 * w.x, sv11 * w.x};
    const float tc[8] = {lo[0], hi[0], lo[2], hi[2], lo[1], hi[1], lo[3], hi[3]};
    static_assert(ORDER == 1 || ORDER == 2, "the wide sum (order 3) is formed corner by corner in k_backward3o itself");
    float acc = wc[0] * tc[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) { const float pr = wc[c] * tc[c]; acc = acc + pr; }
    return acc;
}

// The same for an action whose blend along a is trivial (ACT_SINGLE_A: both a-rows are the same row, or the weight along a is
// exactly 0): lo / hi hold the a0 rows B0 (v0), B2 (v1) only.  With twin rows the a1 taps ARE the a0 taps; with weight 0 they
// only meet the factor 0 (finite table values: the product is +0, the blend returns its other operand) -- so the a1 taps are
// replaced by the a0 taps, and under order 1 the two s-blends and the v-blend of the a1 side are the a0 side's, bit for bit.
template <int ORDER, int LERP>
__device__ __forceinline__ float fetch8_single(const float (&lo)[2], const float (&hi)[2], float ws, float iws, const v4f& w)
{
    if (ORDER == 1) {
        const float x00 = blend1<LERP>(lo[0], hi[0], ws, iws), x01 = blend1<LERP>(lo[1], hi[1], ws, iws);
        const float y0 = blend1<LERP>(x00, x01, w.z, w.w);
        return blend1<LERP>(y0, y0, w.x, w.y);
    }
    // the sums: with a weight of exactly 0 along a (uniform for the action: a scalar test) the four a1 corners carry the weight
    // (w_s w_v) * 0 = 0, their products are +-0 and leave the accumulator as it is, and the a0 corners' weight is (w_s w_v) * 1:
    // four terms instead of eight, no product with w_a.  Twin rows with another weight keep all eight terms on the twin taps.
    // (As with every ACT_SINGLE_A shortcut, the one thing that can differ is the SIGN of a result that is itself a zero -- a -0 sum
    // of four -0 terms would have met +0 products -- and no consumer sees it: eval_action_tail adds the node to sums that start at
    // +0.0f, and squares (node.z - j).)
    if (__builtin_amdgcn_readfirstlane(__float_as_int(w.x)) == 0) {
        const float wc[4] = {iws * w.w, ws * w.w, iws * w.z, ws * w.z};
        const float tc[4] = {lo[0], hi[0], lo[1], hi[1]};
        float acc = wc[0] * tc[0];
#pragma unroll
        for (int c = 1; c < 4; ++c) { const float pr = wc[c] * tc[c]; acc = acc + pr; }
        return acc;
    }
    const float l4[4] = {lo[0], lo[0], lo[1], lo[1]}, h4[4] = {hi[0], hi[0], hi[1], hi[1]};
    return fetch8<ORDER, LERP>(l4, h4, ws, iws, w);
}

// (six wavefronts per SIMD for the sequential order, whose kernels fit 80 registers without a spill: 59.9 -> 55.2 us per
// 256x128x64 sweep; the sums reach six by themselves -- the fp32 sum at 75-80 registers once a zero weight along a drops four
// corners, the wide sum at 67-72 formed corner by corner; their first forms needed 83-96 and, held to 80, spilled and lost)
template <int PATH_DIV, int RANGE_DIV, int QUANT, int LERP, int ORDER>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(ORDER == 1 ? 6 : 1)))
k_backward3o(const Node* __restrict__ next_all, Node* __restrict__ out_all, const ActDesc* __restrict__ acts,
              const float* __restrict__ path_all, const float* __restrict__ pre_vmax_all,
              const float* __restrict__ pre_sdist_all, const double* __restrict__ pre_prog, SweepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TplDpLonParams& p = a.p;
    const int S = p.s_steps, V = p.v_steps, A = p.a_steps;
    const int nrows = V * A;
    const int is = threadIdx.x;
    const int s_pad = a.s_pad;
    const size_t scn = blockIdx.y;
    const Node* __restrict__ next = next_all + scn * a.stride_tables;
    Node* __restrict__ out = out_all + scn * a.stride_tables;
    const float* __restrict__ pre_vmax = pre_vmax_all + scn * a.stride_pre;
    const float* __restrict__ pre_sdist = pre_sdist_all + scn * a.stride_pre;
    const float* __restrict__ path = path_all + scn * a.stride_path;

    constexpr int NV = 2;
    // traversal: v3_map_block (dp_lon_internal.h) -- read from the table the host filled at reinit, or computed
    int ia, iv;
    if (a.tile_map) {
        const unsigned m = a.tile_map[blockIdx.x].where;
        if (m == V3_MAP_NONE) return;
        ia = (int)(m >> 16); iv = (int)(m & 0xffffu);
    } else if (!v3_map_block((int)blockIdx.x, V, A, a.tile_v, a.tile_a, a.skew_q16, ia, iv)) {
        return;
    }
    const int row0 = ia * V + iv;
    const int rs = (iv + 1 < V) ? N_BWD : 0;            // descriptor stride to the lane's second row (0: the row twice)
    const bool active = is < S;
    const int js = min(is, S - 1);

    // LDS: float R[s_pad + 1][SLOT_DW] (rounded up to 16 bytes) | float dsc[NV rows][9][8] | float pv[path_steps + 1]
    const int row_slots = s_pad + 1;
    float* R = reinterpret_cast<float*>(smem);
    float* dsc = reinterpret_cast<float*>(smem + (((size_t)row_slots * SLOT_DW * sizeof(float) + 15) & ~(size_t)15));
    float* pv = dsc + NV * N_BWD * 8;
    const ActDesc* __restrict__ d0 = acts + (size_t)row0 * N_BWD;
    const __amdgpu_buffer_rsrc_t next_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<Node*>(next), 0, (int)((size_t)nrows * S * sizeof(Node)), 0x00020000);
    const unsigned voff = (unsigned)js * 12u, row_bytes = (unsigned)S * 12u;
    auto load_row = [&](int r) -> Node3 {
        const v3f x = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(next_rsrc, (int)voff, (int)((unsigned)r * row_bytes), 0));
        return Node3{x.x, x.y, x.z};
    };
    Node3 lo[2], up[NV][2];
    struct ActScalars { int r00, r10, r01[NV], r11[NV], single, shared; double j_term[NV], tn_v_d[NV]; };
    auto fetch_scalars = [&](int k) {
        ActScalars x;
        x.r00 = d0[k].r00; x.r10 = d0[k].r10;
        const int fl = d0[k].flags;
        x.single = fl & ACT_SINGLE_A;               // (a property of (a, k): the same for both rows of the pair)
        x.shared = fl & ACT_SHARED_UP;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const ActDesc& e = d0[i * rs + k];
            x.r01[i] = e.r01; x.r11[i] = e.r11; x.j_term[i] = e.j_term; x.tn_v_d[i] = e.tn_v_d;
        }
        return x;
    };
    constexpr unsigned A0_ROWS = 0x15u, A1_ROWS = 0x2au;
    static_assert(NV == 2, "row masks are written for two rows per lane");
    auto issue = [&](const ActScalars& x, unsigned m) {
        if (m & 1u) lo[0] = load_row(x.r00);
        if (m & 2u) lo[1] = load_row(x.r10);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if ((m >> (2 + 2 * i)) & 1u) up[i][0] = load_row(x.r01[i]);
            if ((m >> (3 + 2 * i)) & 1u) up[i][1] = load_row(x.r11[i]);
        }
    };
    ActScalars cur = fetch_scalars(0);
    const int bd = (int)blockDim.x;
    auto dsc_word = [&](int i) { return reinterpret_cast<const float*>(d0 + (i / (N_BWD * 8)) * rs + ((i >> 3) % N_BWD))[i & 7]; };
    float dsc_v = 0.0f, pv_v = 0.0f;
    if (is < NV * N_BWD * 8) dsc_v = dsc_word(is);
    if (is <= p.path_steps) pv_v = path[min(is, p.path_steps - 1) * 7 + 5];
    const float s = mad1<LERP>((float)js, p._s_step_size, p.s_min);
    const float acc = mad1<LERP>((float)ia, p._a_step_size, p.a_min);
    const float v_max = pre_vmax[js];
    const float s_dist = pre_sdist[js];
    const double prog = pre_prog[js];
    if (is < NV * N_BWD * 8) dsc[is] = dsc_v;
    if (is <= p.path_steps) pv[is] = pv_v;
    for (int i = is + bd; i < NV * N_BWD * 8; i += bd) dsc[i] = dsc_word(i);
    for (int i = is + bd; i <= p.path_steps; i += bd) pv[i] = path[min(i, p.path_steps - 1) * 7 + 5];
    float sc[NV], sg[NV];
    double sd_base[NV];
    {
        float c0 = 0.0f;
        c0 += p.w_a * (acc * acc);
        const float x = kFma<LERP> ? (float)fma((double)p.w_progress, prog, (double)c0) : (float)((double)c0 + prog);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float v = mad1<LERP>((float)(iv + (rs ? i : 0)), p._v_step_size, p.v_min);
            const float gap = mad1<LERP>(v, p.time_gap, p.gap_min) - s_dist;
            sc[i] = (float)mad1<LERP>((double)p.w_safety_dist, fmax(0.0, (double)gap), (double)x);
            sg[i] = 0.0f; if (v > v_max) sg[i] += v - v_max;
            sd_base[i] = kFma<LERP> ? fma(d0[i * rs].a_term, (double)p.dt, (double)fmaf(d0[i * rs].vdt, p.dt, s))
                                    : (double)(s + d0[i * rs].vdt) + d0[i * rs].a_term;
        }
    }
    const float s_min_v = in_vgpr(p.s_min);
    const float s_scale = (float)(S - 1);
    const float s_scale_v = in_vgpr(s_scale);
    const float rr_hi = in_vgpr(a.rcp_range_hi), rr_lo = in_vgpr(a.rcp_range_lo);
    const float rp_hi = in_vgpr(a.rcp_path_hi), rp_lo = in_vgpr(a.rcp_path_lo);
    const float w_snap_v = in_vgpr(p.w_snap);
    const float path_top = (float)(p.path_steps - 1);
    const double dt_d = (double)p.dt;
    const bool dup_last = is == S - 1;
    unsigned long long best[NV];
    int bk[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { best[i] = 0x7f8000007f800000ull; bk[i] = 0; }
    __syncthreads();                     // descriptors and path limits are in LDS
    const v4f* dsc4 = reinterpret_cast<const v4f*>(dsc);
    float* slot = R + is * SLOT_DW;
    // one row into its place of a slot
    const auto park_row = [&](float* sl, int b, const Node3& r) { sl[b * 3 + 0] = r.c; sl[b * 3 + 1] = r.g; sl[b * 3 + 2] = r.j; };
#pragma unroll
    for (int k = 0; k < N_BWD; ++k) {
        // ---- phase 1: the raw rows of the pair into LDS
        const bool shared = cur.shared != 0;           // ACT_SHARED_UP (k_build_actions)
        Node3 q00 = {0.0f, 0.0f, 0.0f}, q10 = {0.0f, 0.0f, 0.0f};
        __builtin_amdgcn_s_setprio(1);
        // (all six rows here: requesting the a0 rows a phase ahead, as k_backward3v does, costs nine registers across phase 2 and
        // buys nothing in a kernel bound by LDS and VALU throughput -- 63.2 / 64.8 / 77.5 us with, 62.9 / 64.8 / 73.8 without)
        // An action whose blend along a is trivial (a third to a half of all (a, action) pairs, see ACT_SINGLE_A) requests and parks
        // its three a0 rows only -- B0, B2, B4 -- and its cells read four taps per channel pair instead of eight (fetch8_single).
        const bool single = cur.single != 0;
        if (single) {
            issue(cur, A0_ROWS);
            if (!shared) q00 = load_row(d0[rs + k].r00);
            if (k > 0) asm volatile("s_barrier" ::: "memory");  // everyone is done reading the row block of action k-1
            park_row(slot, 0, lo[0]); park_row(slot, 2, up[0][0]); park_row(slot, 4, up[1][0]);
            if (dup_last) { float* d = slot + SLOT_DW; park_row(d, 0, lo[0]); park_row(d, 2, up[0][0]); park_row(d, 4, up[1][0]); }
        } else {
            issue(cur, A0_ROWS | A1_ROWS);
            if (!shared) { q00 = load_row(d0[rs + k].r00); q10 = load_row(d0[rs + k].r10); }
            if (k > 0) asm volatile("s_barrier" ::: "memory");
            park_row(slot, 0, lo[0]); park_row(slot, 1, lo[1]);
            park_row(slot, 2, up[0][0]); park_row(slot, 3, up[0][1]);
            park_row(slot, 4, up[1][0]); park_row(slot, 5, up[1][1]);
            if (dup_last) {                                      // the duplicate of the last station: one branch per action
                float* d = slot + SLOT_DW;
                park_row(d, 0, lo[0]); park_row(d, 1, lo[1]);
                park_row(d, 2, up[0][0]); park_row(d, 3, up[0][1]);
                park_row(d, 4, up[1][0]); park_row(d, 5, up[1][1]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_s_setprio(0);
        // ---- phase 2: successor station, the eight taps, R3 in this ORDER, costs, argmin -- per cell
        int kv = k;
        asm volatile("" : "+v"(kv));
        const ActScalars now = cur;
        if (k + 1 < N_BWD) {
            cur = fetch_scalars(k + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i == 1 && !shared) {
                // cell 1 does not share cell 0's upper rows: its own lower rows take their place
                asm volatile("s_barrier" ::: "memory");
                park_row(slot, 2, q00);
                if (!single) park_row(slot, 3, q10);
                if (dup_last) { park_row(slot + SLOT_DW, 2, q00); if (!single) park_row(slot + SLOT_DW, 3, q10); }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            const v4f w = dsc4[(i * N_BWD + k) * 2];
            const v4f u = dsc4[(i * N_BWD + k) * 2 + 1];
            const double sd = kFma<LERP> ? fma(now.j_term[i], dt_d, sd_base[i]) : sd_base[i] + now.j_term[i];
            const float tn_s = fmaxf(s, (float)sd);
            // (the path division first: its tiny-dividend branch ends a basic block, and whatever is in flight across it -- the
            // 24 taps, were they requested first -- stays in registers on both sides)
            const float alpha = div_path<PATH_DIV>(tn_s, rp_hi, rp_lo, p.path_step_size);
            const float xs = tn_s - s_min_v;
            const float qs = div_const<RANGE_DIV>(xs, rr_hi, rr_lo);
            const float cs = qs * s_scale_v + 0.5f;
            const float xb = cs - 0.5f;
            const float fl = floorf(xb);
            const float fr = xb - fl;
            const float ws = s_weight<QUANT>(fr);
            const float iws = 1.0f - ws;
            const int off = (int)(fminf(fl, s_scale) * (float)(SLOT_DW * 4));
            const float* T0 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(R) + off) + i * 6;
            float nch[3];
            if constexpr (ORDER == 3) {
                // the wide sum corner by corner: one weight as a double at a time, three accumulators (the channel-by-channel form
                // of fetch8 kept eight weights as doubles: 92-96 registers against these 67-72)
                const float sv00 = iws * w.w, sv10 = ws * w.w, sv01 = iws * w.z, sv11 = ws * w.z;
                double acc[3];
                auto term = [&](bool first, float wf, int o) {
                    asm volatile("" : "+v"(wf));
                    const double wd = (double)wf;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) acc[ch] = first ? wd * (double)T0[o + ch] : fma(wd, (double)T0[o + ch], acc[ch]);
                };
                if (single) {
                    // the a0 rows B0 (v0), B2 (v1) only: with a weight of exactly 0 along a the four a0 corners with the weights
                    // w_s w_v (see fetch8_single), otherwise the twin taps under all eight weights
                    if (__builtin_amdgcn_readfirstlane(__float_as_int(w.x)) == 0) {
                        term(true, sv00, 0); term(false, sv10, SLOT_DW); term(false, sv01, 6); term(false, sv11, SLOT_DW + 6);
                    } else {
                        term(true, sv00 * w.y, 0); term(false, sv10 * w.y, SLOT_DW); term(false, sv01 * w.y, 6); term(false, sv11 * w.y, SLOT_DW + 6);
                        term(false, sv00 * w.x, 0); term(false, sv10 * w.x, SLOT_DW); term(false, sv01 * w.x, 6); term(false, sv11 * w.x, SLOT_DW + 6);
                    }
                } else {
                    term(true, sv00 * w.y, 0); term(false, sv10 * w.y, SLOT_DW); term(false, sv01 * w.y, 6); term(false, sv11 * w.y, SLOT_DW + 6);
                    term(false, sv00 * w.x, 3); term(false, sv10 * w.x, SLOT_DW + 3); term(false, sv01 * w.x, 9); term(false, sv11 * w.x, SLOT_DW + 9);
                }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) nch[ch] = (float)acc[ch];
            } else if (single) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float tl[2] = {T0[ch], T0[6 + ch]};
                    const float th[2] = {T0[SLOT_DW + ch], T0[SLOT_DW + 6 + ch]};
                    nch[ch] = fetch8_single<ORDER, LERP & 3>(tl, th, ws, iws, w);
                }
            } else {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float tl[4] = {T0[ch], T0[3 + ch], T0[6 + ch], T0[9 + ch]};
                    const float th[4] = {T0[SLOT_DW + ch], T0[SLOT_DW + 3 + ch], T0[SLOT_DW + 6 + ch], T0[SLOT_DW + 9 + ch]};
                    nch[ch] = fetch8<ORDER, LERP & 3>(tl, th, ws, iws, w);
                }
            }
            const float nc = nch[0], ng = nch[1], nj = nch[2];
            const float f0 = fminf(floorf(alpha), path_top);
            const float pa = alpha - f0;
            const float pia = 1.0f - pa;
            const int t4 = (int)(f0 * 4.0f);
            const float pv0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4);
            const float pv1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4 + 4);
            const float vmn = mad1<LERP>(pv0, pia, pv1 * pa);
            float c = sc[i] + nc;
            float g = sg[i] + ng;
            const float dj = nj - u.x;
            c = mad1<LERP>(w_snap_v, dj * dj, c);
            g = (float)((double)g + fmax(now.tn_v_d[i] - (double)vmn, 0.0));
            const float s_change = tn_s - s;
            g += fmaxf(s_change - s_dist, 0.0f);
            c += u.y;
            g += u.z;
            const unsigned long long key = ((unsigned long long)__float_as_uint(g) << 32) | __float_as_uint(c);
            const bool better = key < best[i];
            best[i] = better ? key : best[i];
            bk[i] = better ? kv : bk[i];
            __builtin_amdgcn_sched_barrier(0);                 // one cell at a time: two cells' taps in flight cost a wavefront per SIMD
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(best[i]), "+v"(bk[i]));
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i == 0 || rs) st_node(out + (size_t)(row0 + i) * S + is, __uint_as_float((unsigned)best[i]), __uint_as_float((unsigned)(best[i] >> 32)),
                    action_jerk(p, bk[i], N_BWD, FilterRule{QUANT, LERP & 3, kFma<LERP> ? 1 : 0, ORDER}));
    }
}

// ------------------------------------------------------------------------
// k_backward3s: the plain-fp32 instruction stream of k_backward3v for grids that do NOT fill the chip (tpl's default
// 201x37x20, 128x64x32): there a sweep is bound by how many instructions ONE SIMD has to issue for its two to four
// wavefronts and by one workgroup's latency, not by loads.  One cell per lane (a row per workgroup: the shortest
// chain per wavefront), KC = 3 actions per barrier round (their row loads requested together), the same arithmetic in
// the same order and the same integer-key argmin as k_backward3v -- k_backward's work at ~2.5 instead of ~4.5 cycles
// per instruction.
// ------------------------------------------------------------------------
template <int PATH_DIV, int RANGE_DIV, int QUANT, int LERP, int KC, bool FIRST_EARLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
k_backward3s(const Node* __restrict__ next_all, Node* __restrict__ out_all, const ActDesc* __restrict__ acts,
              const float* __restrict__ path_all, const float* __restrict__ pre_vmax_all,
              const float* __restrict__ pre_sdist_all, const double* __restrict__ pre_prog, SweepArgs a)
{
    static_assert(N_BWD % KC == 0, "the chunks cover the actions exactly");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TplDpLonParams& p = a.p;
    const int S = p.s_steps, V = p.v_steps, A = p.a_steps;
    const int nrows = V * A;
    const int is = threadIdx.x;
    const int s_pad = a.s_pad;
    const size_t scn = blockIdx.y;
    const Node* __restrict__ next = next_all + scn * a.stride_tables;
    Node* __restrict__ out = out_all + scn * a.stride_tables;
    const float* __restrict__ pre_vmax = pre_vmax_all + scn * a.stride_pre;
    const float* __restrict__ pre_sdist = pre_sdist_all + scn * a.stride_pre;
    const float* __restrict__ path = path_all + scn * a.stride_path;

    // traversal: k_backward's (XCD = blockIdx.x & 7 owns a v range, strips of tile_v rows walked a-major)
    const int xcd = blockIdx.x & 7;
    const int v_lo = (xcd * V) >> 3, v_hi = ((xcd + 1) * V) >> 3;
    const int vx = v_hi - v_lo, tv = max(1, a.tile_v);
    const int q = (int)(blockIdx.x >> 3);
    if (q >= vx * A) return;
    int ia, iv;
    {
        const int full = vx / tv;
        int strip = q / (A * tv), w = tv, rem = q - strip * (A * tv);
        if (strip >= full) { strip = full; w = vx - full * tv; rem = q - full * (A * tv); }
        const int tile = rem / (a.tile_a * w);
        const int r = rem - tile * (a.tile_a * w);
        ia = tile * a.tile_a + r / w;
        iv = v_lo + strip * tv + (r - (r / w) * w);
    }
    const int row0 = ia * V + iv;
    const bool active = is < S;
    const int js = min(is, S - 1);

    // LDS: float4 R[KC actions][s_pad + 1] | float dsc[9][8] | float pv[path_steps + 1]
    const int row_slots = s_pad + 1;
    v4f* R = reinterpret_cast<v4f*>(smem);
    float* dsc = reinterpret_cast<float*>(smem + (size_t)KC * row_slots * sizeof(v4f));
    float* pv = dsc + N_BWD * 8;
    const ActDesc* __restrict__ d0 = acts + (size_t)row0 * N_BWD;
    const __amdgpu_buffer_rsrc_t next_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<Node*>(next), 0, (int)((size_t)nrows * S * sizeof(Node)), 0x00020000);
    const unsigned voff = (unsigned)js * 12u, row_bytes = (unsigned)S * 12u;
    auto load_row = [&](int r) -> Node3 {
        const v3f x = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(next_rsrc, (int)voff, (int)((unsigned)r * row_bytes), 0));
        return Node3{x.x, x.y, x.z};
    };
    // FIRST_EARLY: the rows of the first round are requested before anything else, so that the prologue's own loads and the
    // rows share one memory round trip (110 registers instead of 69).  Same-box A/B: 128x64x32 (4096 wavefronts, four per
    // SIMD) 11.45 -> 10.5 us per sweep; 201x37x20 (2960 wavefronts) 9.89 -> 10.06 us: the host asks for it where the
    // grid fills four wavefronts per SIMD (profiles/r03_dp_sweep_ablation.txt, block 5).
    Node3 r00[KC], r10[KC], r01[KC], r11[KC];
    // (ACT_SINGLE_A, see k_backward3v: an action whose blend along a is trivial requests its two a0 rows only)
    // (not in the FIRST_EARLY instance, where it lost 1-9 % on 128x64x32; 201x37x20 -- the other instance -- gains 3 % per
    // plan: profiles/r03_dp_sweep_ablation.txt, block 9)
    auto request_rows = [&](int kk, const ActDesc& x, bool may_skip) {
        r00[kk] = load_row(x.r00);
        r01[kk] = load_row(x.r01);
        if (!(may_skip && (x.flags & ACT_SINGLE_A))) { r10[kk] = load_row(x.r10); r11[kk] = load_row(x.r11); }
    };
    if constexpr (FIRST_EARLY) {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) request_rows(kk, d0[kk], false);
    }
    // ONE memory round trip for the prologue: every lane requests its share of the descriptor words, of the path limits
    // and its stage prelude values first, then stores to LDS (written as copy loops, each loop waits for its own loads:
    // three round trips in front of the first row load on a grid where a sweep lasts as long as one workgroup's chain)
    // (same-box A/B: 201x37x20 9.88 -> 9.73 us per sweep, 256x128x64 40.38 -> 40.20 in k_backward3v; behind the twelve early
    // row requests of FIRST_EARLY the prologue's small loads only queue up, 10.46 -> 10.75: there the copy loops stay)
    const int bd = FIRST_EARLY ? 0x40000000 : (int)blockDim.x;
    float dsc_v = 0.0f, pv_v = 0.0f;
    if constexpr (FIRST_EARLY) {
        for (int i = is; i < N_BWD * 8; i += blockDim.x) dsc[i] = reinterpret_cast<const float*>(d0 + (i >> 3))[i & 7];
        for (int i = is; i <= p.path_steps; i += blockDim.x) pv[i] = path[min(i, p.path_steps - 1) * 7 + 5];
    }
    if (!FIRST_EARLY && is < N_BWD * 8) dsc_v = reinterpret_cast<const float*>(d0 + (is >> 3))[is & 7];
    if (!FIRST_EARLY && is <= p.path_steps) pv_v = path[min(is, p.path_steps - 1) * 7 + 5];
    const float s = mad1<LERP>((float)js, p._s_step_size, p.s_min);
    const float acc = mad1<LERP>((float)ia, p._a_step_size, p.a_min);
    const float v_max = pre_vmax[js];
    const float s_dist = pre_sdist[js];
    const double prog = pre_prog[js];
    if (!FIRST_EARLY && is < N_BWD * 8) dsc[is] = dsc_v;
    if (!FIRST_EARLY && is <= p.path_steps) pv[is] = pv_v;
    for (int i = is + bd; i < N_BWD * 8; i += bd) dsc[i] = reinterpret_cast<const float*>(d0 + (i >> 3))[i & 7];      // rows of 64 lanes
    for (int i = is + bd; i <= p.path_steps; i += bd) pv[i] = path[min(i, p.path_steps - 1) * 7 + 5];
    float sc, sg;
    double sd_base;
    {
        float c0 = 0.0f;
        c0 += p.w_a * (acc * acc);
        const float x = kFma<LERP> ? (float)fma((double)p.w_progress, prog, (double)c0) : (float)((double)c0 + prog);
        const float v = mad1<LERP>((float)iv, p._v_step_size, p.v_min);
        const float gap = mad1<LERP>(v, p.time_gap, p.gap_min) - s_dist;
        sc = (float)mad1<LERP>((double)p.w_safety_dist, fmax(0.0, (double)gap), (double)x);
        sg = 0.0f; if (v > v_max) sg += v - v_max;
        sd_base = kFma<LERP> ? fma(d0[0].a_term, (double)p.dt, (double)fmaf(d0[0].vdt, p.dt, s)) : (double)(s + d0[0].vdt) + d0[0].a_term;
    }
    const float s_min_v = in_vgpr(p.s_min);
    const float s_scale = (float)(S - 1);
    const float s_scale_v = in_vgpr(s_scale);
    const float rr_hi = in_vgpr(a.rcp_range_hi), rr_lo = in_vgpr(a.rcp_range_lo);
    const float rp_hi = in_vgpr(a.rcp_path_hi), rp_lo = in_vgpr(a.rcp_path_lo);
    const float w_snap_v = in_vgpr(p.w_snap);
    const float path_top = (float)(p.path_steps - 1);
    const bool dup_last = is == S - 1;
    unsigned long long best = 0x7f8000007f800000ull;
    int bk = 0;
    const v4f* dsc4 = reinterpret_cast<const v4f*>(dsc);
    // (k_backward3v's pin of the state terms in front of the first barrier gains nothing here: 10.4-10.8 us per 201x37x20 sweep
    // either way, profiles/r06_dp_contraction_ab.txt)

#pragma unroll
    for (int k0 = 0; k0 < N_BWD; k0 += KC) {
        // ---- phase 1: the four rows of each of the KC actions, requested together; then a-blend, v-blend
        if (k0 > 0 || !FIRST_EARLY) {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) request_rows(kk, d0[k0 + kk], !FIRST_EARLY);
        }
        if (k0 > 0) asm volatile("s_barrier" ::: "memory");          // the row buffers of the last chunk are read
        else __syncthreads();                                         // descriptors and path limits are in LDS
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const v4f w = dsc4[(k0 + kk) * 2];
            const bool single = !FIRST_EARLY && (d0[k0 + kk].flags & ACT_SINGLE_A);
            const Node3 b0 = single ? lerp3<LERP>(r00[kk], r00[kk], w.x, w.y) : lerp3<LERP>(r00[kk], r10[kk], w.x, w.y);
            const Node3 b1 = single ? lerp3<LERP>(r01[kk], r01[kk], w.x, w.y) : lerp3<LERP>(r01[kk], r11[kk], w.x, w.y);
            const Node3 r = lerp3<LERP>(b0, b1, w.z, w.w);
            v4f* Rk = R + kk * row_slots;
            *reinterpret_cast<v3f*>(Rk + is) = (v3f){r.c, r.g, r.j};
            if (dup_last) *reinterpret_cast<v3f*>(Rk + S) = (v3f){r.c, r.g, r.j};
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- phase 2: successor station, 1-D blend along s, costs, argmin (k_backward3v's, one cell)
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const int k = k0 + kk;
            int kv = k;
            asm volatile("" : "+v"(kv));
            const ActDesc& e = d0[k];
            const v4f u = dsc4[k * 2 + 1];
            const double sd = kFma<LERP> ? fma(e.j_term, (double)p.dt, sd_base) : sd_base + e.j_term;      // lonDynamics station part, lon_planner.cu:244
            const float tn_s = fmaxf(s, (float)sd);
            const float xs = tn_s - s_min_v;
            const float qs = div_const<RANGE_DIV>(xs, rr_hi, rr_lo);
            const float cs = qs * s_scale_v + 0.5f;
            const float xb = cs - 0.5f;
            const float fl = floorf(xb);
            const float fr = xb - fl;
            const float ws = s_weight<QUANT>(fr);
            const float iws = 1.0f - ws;
            const v4f* Rk = R + kk * row_slots;
            const int off = (int)(fminf(fl, s_scale) * 16.0f);
            const v4f nl = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(Rk) + off);
            const v4f nh = *reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(Rk) + off + 16);
            asm volatile("" :: "v"(nl.w), "v"(nh.w));
            const float nc = blend1<LERP>(nl.x, nh.x, ws, iws);
            const float ng = blend1<LERP>(nl.y, nh.y, ws, iws);
            const float nj = blend1<LERP>(nl.z, nh.z, ws, iws);
            const float alpha = div_path<PATH_DIV>(tn_s, rp_hi, rp_lo, p.path_step_size);
            const float f0 = fminf(floorf(alpha), path_top);
            const float pa = alpha - f0;
            const float pia = 1.0f - pa;
            const int t4 = (int)(f0 * 4.0f);
            const float pv0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4);
            const float pv1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pv) + t4 + 4);
            const float vmn = mad1<LERP>(pv0, pia, pv1 * pa);
            float c = sc + nc;
            float g = sg + ng;
            const float dj = nj - u.x;
            c = mad1<LERP>(w_snap_v, dj * dj, c);
            g = (float)((double)g + fmax(e.tn_v_d - (double)vmn, 0.0));
            const float s_change = tn_s - s;
            g += fmaxf(s_change - s_dist, 0.0f);
            c += u.y;
            g += u.z;
            const unsigned long long key = ((unsigned long long)__float_as_uint(g) << 32) | __float_as_uint(c);
            const bool better = key < best;
            best = better ? key : best;
            bk = better ? kv : bk;
            asm volatile("" : "+v"(best), "+v"(bk));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (active)
        st_node(out + (size_t)row0 * S + is, __uint_as_float((unsigned)best), __uint_as_float((unsigned)(best >> 32)),
                action_jerk(p, bk, N_BWD, FilterRule{QUANT, LERP & 3, kFma<LERP> ? 1 : 0}));
}

} // namespace

// ---- launchers.  This file is compiled ten times (Makefile): TPL_SWEEP_RULESET 0 = the default rule (R1 = 1 contracted,
// R2 rint, R3 fma form) with every pair of division forms, 9 = the same filter rule under R1 = 0 (one rounding per
// operation: the default until round 5), also with every pair; 1 and 2 = four of the eight other (R2, R3) rules each under
// R1 = 0, 3 and 4 = those eight under R1 = 1, all of those with both divisions in the Newton form (5, 5) or both as IEEE
// divisions (0, 0) -- such a rule is only ever selected to follow a dump of the reference build, and then runs on the same
// plain-fp32 sweeps as the default; 5 - 8 = the ordered sweeps k_backward3o.
#ifndef TPL_SWEEP_RULESET
#define TPL_SWEEP_RULESET 0
#endif

namespace {

template <int QUANT, int LERP, bool CHAIN, class Go>
bool pick_div_v(int path_div, int range_div, bool all_forms, Go go)
{
    if (path_div == 5 && range_div == 5) return go(k_backward3v<5, 5, QUANT, LERP, CHAIN>);
    if (path_div == 0 && range_div == 0) return go(k_backward3v<0, 0, QUANT, LERP, CHAIN>);
    if constexpr (QUANT == 1 && (LERP & 3) == 0) {
        if (all_forms) {
            if (path_div == 4 && range_div == 3) return go(k_backward3v<4, 3, QUANT, LERP, CHAIN>);
            if (path_div == 3 && range_div == 3) return go(k_backward3v<3, 3, QUANT, LERP, CHAIN>);
            if (path_div == 4 && range_div == 4) return go(k_backward3v<4, 4, QUANT, LERP, CHAIN>);
            if (path_div == 3 && range_div == 4) return go(k_backward3v<3, 4, QUANT, LERP, CHAIN>);
            if (path_div == 4 && range_div == 5) return go(k_backward3v<4, 5, QUANT, LERP, CHAIN>);
            if (path_div == 3 && range_div == 5) return go(k_backward3v<3, 5, QUANT, LERP, CHAIN>);
            if (path_div == 5 && range_div == 3) return go(k_backward3v<5, 3, QUANT, LERP, CHAIN>);
            if (path_div == 5 && range_div == 4) return go(k_backward3v<5, 4, QUANT, LERP, CHAIN>);
            if (path_div == 4 && range_div == 0) return go(k_backward3v<4, 0, QUANT, LERP, CHAIN>);
            if (path_div == 3 && range_div == 0) return go(k_backward3v<3, 0, QUANT, LERP, CHAIN>);
            if (path_div == 5 && range_div == 0) return go(k_backward3v<5, 0, QUANT, LERP, CHAIN>);
            if (path_div == 0 && range_div == 3) return go(k_backward3v<0, 3, QUANT, LERP, CHAIN>);
            if (path_div == 0 && range_div == 4) return go(k_backward3v<0, 4, QUANT, LERP, CHAIN>);
            if (path_div == 0 && range_div == 5) return go(k_backward3v<0, 5, QUANT, LERP, CHAIN>);
        }
    }
    return false;
}

template <int QUANT, int LERP, bool FE, class Go>
bool pick_div_s(int path_div, int range_div, bool all_forms, Go go)
{
    if (path_div == 5 && range_div == 5) return go(k_backward3s<5, 5, QUANT, LERP, 3, FE>);
    if (path_div == 0 && range_div == 0) return go(k_backward3s<0, 0, QUANT, LERP, 3, FE>);
    if constexpr (QUANT == 1 && (LERP & 3) == 0) {
        if (all_forms) {
            if (path_div == 4 && range_div == 3) return go(k_backward3s<4, 3, QUANT, LERP, 3, FE>);
            if (path_div == 3 && range_div == 3) return go(k_backward3s<3, 3, QUANT, LERP, 3, FE>);
            if (path_div == 4 && range_div == 4) return go(k_backward3s<4, 4, QUANT, LERP, 3, FE>);
            if (path_div == 3 && range_div == 4) return go(k_backward3s<3, 4, QUANT, LERP, 3, FE>);
            if (path_div == 4 && range_div == 5) return go(k_backward3s<4, 5, QUANT, LERP, 3, FE>);
            if (path_div == 3 && range_div == 5) return go(k_backward3s<3, 5, QUANT, LERP, 3, FE>);
            if (path_div == 5 && range_div == 3) return go(k_backward3s<5, 3, QUANT, LERP, 3, FE>);
            if (path_div == 5 && range_div == 4) return go(k_backward3s<5, 4, QUANT, LERP, 3, FE>);
            if (path_div == 4 && range_div == 0) return go(k_backward3s<4, 0, QUANT, LERP, 3, FE>);
            if (path_div == 3 && range_div == 0) return go(k_backward3s<3, 0, QUANT, LERP, 3, FE>);
            if (path_div == 5 && range_div == 0) return go(k_backward3s<5, 0, QUANT, LERP, 3, FE>);
            if (path_div == 0 && range_div == 3) return go(k_backward3s<0, 3, QUANT, LERP, 3, FE>);
            if (path_div == 0 && range_div == 4) return go(k_backward3s<0, 4, QUANT, LERP, 3, FE>);
            if (path_div == 0 && range_div == 5) return go(k_backward3s<0, 5, QUANT, LERP, 3, FE>);
        }
    }
    return false;
}

template <int QUANT, int LERP>
bool launch_v_rule(int path_div, int range_div, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    auto go = [&](auto kernel) {
        if (t_sweep_probe) return true;
        hipLaunchKernelGGL(kernel, grid, dim3(a.s_pad), lds_bytes, stream, a.next, a.out, a.acts, a.path, a.pre_vmax, a.pre_sdist, a.pre_prog, a);
        return true;
    };
    if (a.chain.stages != 0) {
        // (the chained form exists for the default filter rule only: dp_lon.hip asks for it under nothing else)
        if constexpr (QUANT == 1 && (LERP & 3) == 0) return pick_div_v<QUANT, LERP, true>(path_div, range_div, true, go);
        else return false;
    }
    return pick_div_v<QUANT, LERP, false>(path_div, range_div, true, go);
}

template <int QUANT, int LERP>
bool launch_s_rule(int path_div, int range_div, bool first_early, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    auto go = [&](auto kernel) {
        if (t_sweep_probe) return true;
        hipLaunchKernelGGL(kernel, grid, dim3(a.s_pad), lds_bytes, stream, a.next, a.out, a.acts, a.path, a.pre_vmax, a.pre_sdist, a.pre_prog, a);
        return true;
    };
    return first_early ? pick_div_s<QUANT, LERP, true>(path_div, range_div, true, go)
                       : pick_div_s<QUANT, LERP, false>(path_div, range_div, true, go);
}

template <int QUANT, int LERP, int ORDER>
bool launch_o_rule(int path_div, int range_div, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    auto go = [&](auto kernel) {
        if (t_sweep_probe) return true;
        hipLaunchKernelGGL(kernel, grid, dim3(a.s_pad), lds_bytes, stream, a.next, a.out, a.acts, a.path, a.pre_vmax, a.pre_sdist, a.pre_prog, a);
        return true;
    };
    if (path_div == 5 && range_div == 5) return go(k_backward3o<5, 5, QUANT, LERP, ORDER>);
    if (path_div == 0 && range_div == 0) return go(k_backward3o<0, 0, QUANT, LERP, ORDER>);
    return false;
}

} // namespace

// (Q, L, F): weight rule, blend form, contraction; the kernels take L + 4 F as their LERP argument
#define TPL_RULE_V(Q, L, F) if (r.quant == Q && r.lerp == L && r.fma == F) return launch_v_rule<Q, L + 4 * F>(pd, rd, a, grid, lds, st);
#define TPL_RULE_S(Q, L, F) if (r.quant == Q && r.lerp == L && r.fma == F) return launch_s_rule<Q, L + 4 * F>(pd, rd, fe, a, grid, lds, st);
// (Q, L, F, O): ... and the blend order / form (k_backward3o); the eight-corner sums (O = 2, 3) have no blend form: L = 0 stands for all
#define TPL_RULE_O(Q, L, F, O) if (r.quant == Q && (O != 1 || r.lerp == L) && r.fma == F && r.order == O) return launch_o_rule<Q, L + 4 * F, O>(pd, rd, a, grid, lds, st);
#define TPL_ORDER_FILE(NAME, RULES_O) \
    bool launch_backward3o_rules_##NAME(FilterRule r, int pd, int rd, const SweepArgs& a, dim3 grid, size_t lds, hipStream_t st) { RULES_O return false; }
#define TPL_RULE_FILE(NAME, RULES_V, RULES_S) \
    bool launch_backward3v_rules_##NAME(FilterRule r, int pd, int rd, const SweepArgs& a, dim3 grid, size_t lds, hipStream_t st) { RULES_V return false; } \
    bool launch_backward3s_rules_##NAME(FilterRule r, int pd, int rd, bool fe, const SweepArgs& a, dim3 grid, size_t lds, hipStream_t st) { RULES_S return false; }
#if TPL_SWEEP_RULESET == 1
TPL_RULE_FILE(a, TPL_RULE_V(1, 1, 0) TPL_RULE_V(1, 2, 0) TPL_RULE_V(2, 0, 0) TPL_RULE_V(2, 1, 0),
                 TPL_RULE_S(1, 1, 0) TPL_RULE_S(1, 2, 0) TPL_RULE_S(2, 0, 0) TPL_RULE_S(2, 1, 0))
#elif TPL_SWEEP_RULESET == 2
TPL_RULE_FILE(b, TPL_RULE_V(2, 2, 0) TPL_RULE_V(0, 0, 0) TPL_RULE_V(0, 1, 0) TPL_RULE_V(0, 2, 0),
                 TPL_RULE_S(2, 2, 0) TPL_RULE_S(0, 0, 0) TPL_RULE_S(0, 1, 0) TPL_RULE_S(0, 2, 0))
#elif TPL_SWEEP_RULESET == 3      /* contracted arithmetic (rule R1 = 1), first half */
TPL_RULE_FILE(c, TPL_RULE_V(2, 0, 1) TPL_RULE_V(0, 0, 1) TPL_RULE_V(1, 1, 1) TPL_RULE_V(1, 2, 1),
                 TPL_RULE_S(2, 0, 1) TPL_RULE_S(0, 0, 1) TPL_RULE_S(1, 1, 1) TPL_RULE_S(1, 2, 1))
#elif TPL_SWEEP_RULESET == 4      /* contracted arithmetic, second half */
TPL_RULE_FILE(d, TPL_RULE_V(2, 1, 1) TPL_RULE_V(2, 2, 1) TPL_RULE_V(0, 1, 1) TPL_RULE_V(0, 2, 1),
                 TPL_RULE_S(2, 1, 1) TPL_RULE_S(2, 2, 1) TPL_RULE_S(0, 1, 1) TPL_RULE_S(0, 2, 1))
#elif TPL_SWEEP_RULESET == 5      /* blend order s -> v -> a (k_backward3o), one rounding per operation */
TPL_ORDER_FILE(e, TPL_RULE_O(1, 0, 0, 1) TPL_RULE_O(1, 1, 0, 1) TPL_RULE_O(1, 2, 0, 1) TPL_RULE_O(2, 0, 0, 1) TPL_RULE_O(2, 1, 0, 1) TPL_RULE_O(2, 2, 0, 1))
#elif TPL_SWEEP_RULESET == 6      /* blend order s -> v -> a, contracted arithmetic */
TPL_ORDER_FILE(f, TPL_RULE_O(1, 0, 1, 1) TPL_RULE_O(1, 1, 1, 1) TPL_RULE_O(1, 2, 1, 1) TPL_RULE_O(2, 0, 1, 1) TPL_RULE_O(2, 1, 1, 1) TPL_RULE_O(2, 2, 1, 1))
#elif TPL_SWEEP_RULESET == 7      /* the eight-corner sum, fp32 accumulator */
TPL_ORDER_FILE(g, TPL_RULE_O(1, 0, 0, 2) TPL_RULE_O(2, 0, 0, 2) TPL_RULE_O(1, 0, 1, 2) TPL_RULE_O(2, 0, 1, 2))
#elif TPL_SWEEP_RULESET == 8      /* the eight-corner sum, wide accumulator */
TPL_ORDER_FILE(h, TPL_RULE_O(1, 0, 0, 3) TPL_RULE_O(2, 0, 0, 3) TPL_RULE_O(1, 0, 1, 3) TPL_RULE_O(2, 0, 1, 3))
#elif TPL_SWEEP_RULESET == 9      /* the default filter rule under R1 = 0 (one rounding per operation), every division form */
TPL_RULE_FILE(i, TPL_RULE_V(1, 0, 0), TPL_RULE_S(1, 0, 0))
#else
#define TPL_ORDER_DECL(NAME) \
    bool launch_backward3o_rules_##NAME(FilterRule r, int pd, int rd, const SweepArgs& a, dim3 grid, size_t lds, hipStream_t st);
TPL_ORDER_DECL(e) TPL_ORDER_DECL(f) TPL_ORDER_DECL(g) TPL_ORDER_DECL(h)
#define TPL_RULE_DECL(NAME) \
    bool launch_backward3v_rules_##NAME(FilterRule r, int pd, int rd, const SweepArgs& a, dim3 grid, size_t lds, hipStream_t st); \
    bool launch_backward3s_rules_##NAME(FilterRule r, int pd, int rd, bool fe, const SweepArgs& a, dim3 grid, size_t lds, hipStream_t st);
TPL_RULE_DECL(a) TPL_RULE_DECL(b) TPL_RULE_DECL(c) TPL_RULE_DECL(d) TPL_RULE_DECL(i)

size_t backward3v_lds_bytes(int s_pad, int path_steps)
{
    return (size_t)2 * (s_pad + 1) * 16 + (size_t)2 * N_BWD * 8 * sizeof(float) + sizeof(float) * ((size_t)path_steps + 1);
}

size_t backward3o_lds_bytes(int s_pad, int path_steps)
{
    return ((((size_t)s_pad + 1) * SLOT_DW * sizeof(float) + 15) & ~(size_t)15) + (size_t)2 * N_BWD * 8 * sizeof(float) + sizeof(float) * ((size_t)path_steps + 1);
}

bool backward3o_built(FilterRule rule)
{
    return rule.order >= 1 && rule.order <= 3 && (rule.quant == 1 || rule.quant == 2);
}

bool launch_backward3v(FilterRule rule, int path_div, int range_div, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    if (rule.order != 0) {                      // another blend order / the eight-corner sums: k_backward3o, its own LDS block
        const size_t lds = backward3o_lds_bytes(a.s_pad, a.p.path_steps);
        return launch_backward3o_rules_e(rule, path_div, range_div, a, grid, lds, stream)
               || launch_backward3o_rules_f(rule, path_div, range_div, a, grid, lds, stream)
               || launch_backward3o_rules_g(rule, path_div, range_div, a, grid, lds, stream)
               || launch_backward3o_rules_h(rule, path_div, range_div, a, grid, lds, stream);
    }
    if (rule.quant == kDefaultRule.quant && rule.lerp == kDefaultRule.lerp)
        return rule.fma ? launch_v_rule<1, 4>(path_div, range_div, a, grid, lds_bytes, stream)
                        : launch_backward3v_rules_i(rule, path_div, range_div, a, grid, lds_bytes, stream);
    if (rule.fma)
        return launch_backward3v_rules_c(rule, path_div, range_div, a, grid, lds_bytes, stream)
               || launch_backward3v_rules_d(rule, path_div, range_div, a, grid, lds_bytes, stream);
    return launch_backward3v_rules_a(rule, path_div, range_div, a, grid, lds_bytes, stream)
           || launch_backward3v_rules_b(rule, path_div, range_div, a, grid, lds_bytes, stream);
}

size_t backward3s_lds_bytes(int kc, int s_pad, int path_steps)
{
    return (size_t)kc * (s_pad + 1) * 16 + (size_t)N_BWD * 8 * sizeof(float) + sizeof(float) * ((size_t)path_steps + 1);
}

// one row per workgroup of a.s_pad <= 256 lanes, grid = 8 * (rows of the widest XCD share); kc = 3 (all nine actions per round
// was built too: 128 registers with spills, 0.22 against 0.16 ms per default-grid plan)
bool launch_backward3s(FilterRule rule, int path_div, int range_div, int kc, bool first_early, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    if (a.s_pad > 256 || lds_bytes > 64 * 1024 || kc != 3) return false;
    if (rule.quant == kDefaultRule.quant && rule.lerp == kDefaultRule.lerp)
        return rule.fma ? launch_s_rule<1, 4>(path_div, range_div, first_early, a, grid, lds_bytes, stream)
                        : launch_backward3s_rules_i(rule, path_div, range_div, first_early, a, grid, lds_bytes, stream);
    if (rule.fma)
        return launch_backward3s_rules_c(rule, path_div, range_div, first_early, a, grid, lds_bytes, stream)
               || launch_backward3s_rules_d(rule, path_div, range_div, first_early, a, grid, lds_bytes, stream);
    return launch_backward3s_rules_a(rule, path_div, range_div, first_early, a, grid, lds_bytes, stream)
           || launch_backward3s_rules_b(rule, path_div, range_div, first_early, a, grid, lds_bytes, stream);
}
#endif

} // namespace tplhip
