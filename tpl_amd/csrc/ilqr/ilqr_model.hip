// ilqr_model.hip -- HP-2: batched augmented-Lagrangian iLQR for ONE generated
// model, hand-written for gfx950.  Compiled once per optimal-control problem
// with -DTPL_MODEL_INC="<problem>.inc" into models/libilqr_<problem>.so; the
// C-ABI is include/tplhip_ilqr.h.
//
// Replaces the solver half of the reference's per-problem C module
// (library/tpl/optim/templates/optim.c:243-291, 657-1177).  MI355X-first design,
// not a translation:
//
//  * ONE WAVEFRONT PER PROBLEM (64-thread workgroup), `batch` workgroups per
//    launch, one launch per update() -- the whole outer loop (initial rollout,
//    multiplier update, every iLQR iteration, line search) runs inside the
//    kernel, so there is no host round trip per iteration.  Two kernels do
//    that: k_ilqr_update (one body, 480 registers, one wavefront per SIMD:
//    what a caller with up to 4 x CUs problems gets) and k_ilqr_fused (the
//    phases as real calls, each compiled to 256 registers: two wavefronts per
//    SIMD hide each other's dependent chains -- every uniform batch beyond
//    4 x CUs problems, whatever the model: pick_form()).  The split sequence
//    k_split_* runs the same phases as launches (the A/B leg).  All three
//    give the same bits; the pieces they share are functions further down
//    (riccati_backward, line_search_rollouts, iteration_tail, ...);
//  * derivative refresh and the multiplier update are parallel over the
//    horizon (lane = stage);
//  * the backward Riccati recursion is sequential in the stage but parallel
//    over matrix entries: the stage's blocks are staged in LDS, lanes own
//    entries of  W = Vxx [fx fu],  Q = L + [fx fu]^T W,  the gains and the new
//    Vxx/Vx (four LDS exchanges per stage).  The matrices are <= 8x8, far too
//    small and too latency-bound for MFMA (v_mfma_f64_16x16x4 would be >90 %
//    padding); DESIGN.md has the numbers;
//  * the eight line-search candidates alpha = 10^-i are rolled out
//    CONCURRENTLY on eight lane quads and the first improving one in the order
//    i = 0..7 is taken, which is exactly the reference's accept-first result
//    (optim.c:859-873) at the cost of a single rollout; their stage costs are
//    formed in that order too (alpha = 1 first, the rest only if it fails);
//  * all state lives in HBM in the layout of the host mirrors (row-major
//    [stage][...] per problem), so upload/download are plain copies and the
//    backward pass reads each stage's blocks as contiguous, coalesced runs.
// Arithmetic is fp64 like the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "builtins.h"
#include "../tplhip_debug.h"
#define TPL_QUAD_ROUTINES 1       /* the model file also emits the quad-cooperative ctDynamicsQ */
#include TPL_MODEL_INC
#include "tplhip_ilqr.h"

namespace {

constexpr int NX = TPL_X;
constexpr int NU = TPL_U;
constexpr int NC = TPL_C;
constexpr int NCM = TPL_CMAX;          // storage width of a multiplier row (>= 1)
constexpr int NXU = NX + NU;
constexpr int N_ALPHA = 8;             // optim.c:861
constexpr int WAVE = 64;

enum { EULER = 0, HEUN = 1, RK4 = 2 };

// widths (doubles per row) of the per-stage fields, in TplIlqrField order
__host__ __device__ constexpr int field_width(int f)
{
    switch (f) {
    case TPL_F_INT_STEP: case TPL_F_NEXT_INT_STEP: return 1;
    case TPL_F_X: case TPL_F_NEXT_X: case TPL_F_PREV_X: case TPL_F_LX: return NX;
    case TPL_F_U: case TPL_F_NEXT_U: case TPL_F_PREV_U: case TPL_F_PREV_K: case TPL_F_LU:
    case TPL_F_G: case TPL_F_K_FF: case TPL_F_U_MAX: case TPL_F_U_MIN: return NU;
    case TPL_F_FX: case TPL_F_LXX: return NX * NX;
    case TPL_F_FU: case TPL_F_LUX: case TPL_F_K_FB: return NX * NU;
    case TPL_F_LUU: return NU * NU;
    case TPL_F_LAGRANGE_MULTIPLIER: return NCM;
    case TPL_F_BARRIER_WEIGHT: case TPL_F_LG_MULT_LIMIT: return NCM;   // one row
    }
    return 0;
}

__host__ __device__ constexpr bool field_is_per_stage(int f)
{
    return f != TPL_F_BARRIER_WEIGHT && f != TPL_F_LG_MULT_LIMIT;
}

#ifdef TPL_ILQR_STAMP      /* diagnostic build only: per-phase cycle sums, never in the shipped library */
#define STAMP(slot) do { if (lane == 0) { const long long now__ = (long long)__builtin_amdgcn_s_memtime(); \
                         stamps[slot] += now__ - t_last; t_last = now__; } } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

struct DevState {
    double* f[TPL_F_COUNT];            // device base of every field, [batch][rows][width]
    double* scalars;                   // [batch][TPL_S_COUNT]
    const Params* params;              // [batch]
    double* cand_x;                    // [batch][N_ALPHA][rows][NX]   line-search candidates
    double* cand_u;                    // [batch][N_ALPHA][rows][NU]
    double* cand_c;                    // [batch][N_ALPHA][rows]       stage costs of the candidates
    int rows;                          // t_cap + 1
    long long* stamps;                 // diagnostic builds only
    int* run;                          // [batch] split-launch form: 1 = the problem left its iteration loop in this round
    // LDS residency (decided by the host per launch, see plan_lds):
    int lds_traj;                      // 1: x, u, k, K, uMax, uMin of the problem are mirrored in LDS
    int n_lds_arrays;                  // array parameters copied into LDS
    int roll_direct;                   // 1: the rollouts read nothing from global memory (mirrors AND every array parameter are in LDS): their stage
                                       // results go straight to the candidate arrays instead of through the LDS ring (line_search_rollouts)
    const double* lds_src[TPL_N_PARAMS + 1];     // global source of array slot i (problem 0)
    long long lds_src_stride[TPL_N_PARAMS + 1];  // doubles between problems (0: shared)
    int lds_len[TPL_N_PARAMS + 1];               // doubles
    int lds_off[TPL_N_PARAMS + 1];               // doubles from the dynamic LDS base
    unsigned long long lds_dyn_flat;   // flat address the host assumed for the dynamic LDS base
    unsigned long long lds_base_flat;  // split / fused kernels: flat address of LDS offset 0 (their phases are functions and build their LDS pointers from it)
};

__device__ __forceinline__ double* fld(const DevState& st, int f, int b)
{
    const int64_t rows = field_is_per_stage(f) ? st.rows : 1;
    return st.f[f] + (int64_t)b * rows * field_width(f);
}

// dynamics(), optim.c:657-730
__device__ __forceinline__ void dynamics(const Params& P, const double* x, const double* u, int t,
                                         double dt, int int_type, double* res)
{
    double k1[NX];
    ctDynamics(P, x, u, (double)t, dt, k1, 1);
    if (int_type == EULER) {
#pragma unroll
        for (int i = 0; i < NX; ++i) res[i] = x[i] + k1[i] * dt;
    } else if (int_type == HEUN) {
        double tmp[NX], k2[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k1[i] * dt;
        ctDynamics(P, tmp, u, (double)t, dt, k2, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) res[i] = x[i] + (k1[i] + k2[i]) * (dt / 2.0);
    } else {
        double tmp[NX], k2[NX], k3[NX], k4[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k1[i] * (dt / 2.0);
        ctDynamics(P, tmp, u, (double)t, dt, k2, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k2[i] * (dt / 2.0);
        ctDynamics(P, tmp, u, (double)t, dt, k3, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k3[i] * dt;
        ctDynamics(P, tmp, u, (double)t, dt, k4, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double acc = k1[i] + 0.0;
            acc = k2[i] * 2.0 + acc;
            acc = k3[i] * 2.0 + acc;
            acc = k4[i] + acc;
            res[i] = x[i] + acc * (dt / 6.0);
        }
    }
}

// dynamics() for a quad of lanes that all hold the same (x, u): the model's quad-cooperative
// ctDynamicsQ spreads its independent look-ups and sin/cos evaluations over the four lanes.  Every
// lane gets the full result.  Models without such structure run the plain routine on every lane.
__device__ __forceinline__ void dynamics_quad(const Params& P, const double* x, const double* u, int t,
                                              double dt, int int_type, double* res)
{
#if defined(TPL_HAS_QUAD_DYNAMICS)
    double k1[NX];
    ctDynamicsQ(P, x, u, (double)t, dt, k1, 1);
    if (int_type == EULER) {
#pragma unroll
        for (int i = 0; i < NX; ++i) res[i] = x[i] + k1[i] * dt;
    } else if (int_type == HEUN) {
        double tmp[NX], k2[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k1[i] * dt;
        ctDynamicsQ(P, tmp, u, (double)t, dt, k2, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) res[i] = x[i] + (k1[i] + k2[i]) * (dt / 2.0);
    } else {
        double tmp[NX], k2[NX], k3[NX], k4[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k1[i] * (dt / 2.0);
        ctDynamicsQ(P, tmp, u, (double)t, dt, k2, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k2[i] * (dt / 2.0);
        ctDynamicsQ(P, tmp, u, (double)t, dt, k3, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k3[i] * dt;
        ctDynamicsQ(P, tmp, u, (double)t, dt, k4, 1);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            double acc = k1[i] + 0.0;
            acc = k2[i] * 2.0 + acc;
            acc = k3[i] * 2.0 + acc;
            acc = k4[i] + acc;
            res[i] = x[i] + acc * (dt / 6.0);
        }
    }
#else
    dynamics(P, x, u, t, dt, int_type, res);
#endif
}

// The workgroup IS one wavefront.  LDS operations of a wave are issued and completed in
// program order, so an exchange through LDS between its lanes needs no s_barrier, and --
// unlike __syncthreads(), which also drains vmcnt -- must not wait for the global loads
// that were issued on purpose to fly across it.  It needs no s_waitcnt either: a ds_read
// behind a ds_write of the same wavefront is executed behind it and returns what was
// written, whichever lane wrote it (until round 5 an lgkmcnt(0) sat here: the reads of an
// exchange were only ISSUED once the writes had completed -- one LDS latency per exchange,
// three exchanges per Riccati stage: 6.94 -> 7.06 k batched iterations/s at 1 024 x N = 50,
// every field of a 512-problem batch bit-identical; profiles/r05_ilqr_guards_ab.txt).  What
// remains is the compiler: it must not move memory operations across.
__device__ __forceinline__ void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Dot product of N terms read from LDS with strides sa / sb.  All 2N reads are issued
// before the first multiply (sched_barrier): hipcc otherwise waits for the LDS after
// every pair of terms, four round trips per product instead of one.
template <int N, class PA, class PB>
__device__ __forceinline__ double dot_lds(PA a, int sa, PB b, int sb)
{
    double x[N], y[N];
#pragma unroll
    for (int m = 0; m < N; ++m) { x[m] = a[m * sa]; y[m] = b[m * sb]; }
    __builtin_amdgcn_sched_barrier(0);
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < N; ++m) acc = fma(x[m], y[m], acc);
    return acc;
}

constexpr int ROLL_RING = 8;
// Layout of the rollout ring.  The eight quads write the same (stage, entry) slot of their candidates in ONE instruction: with a
// block per candidate -- ring_x[cand][stage][entry], blocks of ROLL_RING * NX = 56 doubles = 448 B, i.e. 64 B modulo the 128 B
// of an LDS bank row -- four of the eight addresses fall on the same bank pair, and with ring_u's blocks of 16 doubles (128 B)
// all eight do: half of the kernel's SQ_LDS_BANK_CONFLICT cycles (tools/micro/lds_conflict_probe.hip: a 64-bit access of
// consecutive, broadcast or 7- / 9-strided doubles has NO conflict on gfx950, 16-byte-strided ones stall 8 cycles per
// instruction; profiles/r06_ilqr_lds_conflicts.txt).  TRANSPOSED -- ring_x[stage][entry][cand], the candidate fastest -- the
// eight addresses are 64 consecutive bytes, the flush reads the ring linearly (lane e: entry e / 8, candidate e % 8: a shift
// and a mask instead of a division by 56) and a candidate's run in global memory is written 64 B at a time.  Same size: padding
// the blocks instead (57 / 17 doubles) costs the fused form its trajectory mirrors -- the MPC's 1 479 + 500 doubles fit a
// workgroup's 20 KB share with three doubles to spare -- and 10 % at 4 096 problems.  TPL_ILQR_RING_T=0: the block layout (A/B).
#ifndef TPL_ILQR_RING_T
#define TPL_ILQR_RING_T 1
#endif
__device__ __forceinline__ int ring_ix(int cand, int k, int i, int n_entry)       // slot of (candidate, stage k of the ring, entry i)
{
    return TPL_ILQR_RING_T ? (k * n_entry + i) * N_ALPHA + cand : (cand * ROLL_RING + k) * n_entry + i;
}

// LDS working set of one problem.  The Riccati pass and the line search never run at the same time (the terminal value
// function is written after the last read of the cost block, the ring is first written after the last gain is stored):
// their blocks share storage, which is what lets a problem's trajectory mirrors fit next to it when two workgroups
// per SIMD are resident (k_split_chain: 20 KB per workgroup).
struct Lds {
    union {
        struct {               // backward pass
            double Vxx[NX * NX];
            double Vx[NX];
            double F[NX * NXU];        // [fx | fu], row-major over the NXU columns
            double W[NX * (NXU + 1)];  // Vxx [fx fu], and Vx as one more column (second-order pass)
            double q[NXU];             // [Qx ; Qu]
            double Qxx[NX * NX];
            double Qux[NU * NX];
            double Quu[NU * NU];
        };
        struct {               // line search
            // rollouts: RING stages of every candidate's (x_{t+1}, u_t) collect here and go out to the candidate
            // arrays in one coalesced sweep of all 64 lanes (see line_search_rollouts); then the stage costs
            double ring_x[N_ALPHA * ROLL_RING * NX];
            double ring_u[N_ALPHA * ROLL_RING * NU];
        };
    };
    double dump[2];            // where lanes without an entry park what they loaded
};

// Diagnostic builds that PRICE a part by leaving it out (results are not solutions; tools/ilqr_ablate.sh,
// tools/ab_ilqr_remap.py): 1 = no backward pass, 2 = no rollouts (k_ilqr_update only); in every launch form: 3 = no stage
// costs of the candidates, 4 = the rollouts cover half the stages -- both change which steps are accepted and with it how
// often the derivatives are refreshed, so they only bound; the clean prices come from DOING A PART TWICE, which leaves every
// result and every decision as it is: 5 = the first half of the stages is rolled out a second time (the time it adds is what
// two problems per wavefront in the rollouts could save per problem at unchanged occupancy and LDS residency), 6 = every
// stage cost of the candidates is evaluated twice (what folding the cost pass into the rollouts could save at most).
#ifndef TPL_ILQR_ABLATE
#define TPL_ILQR_ABLATE 0
#endif
// 1: the gain solve's shared scalars by v_readlane from the lanes that formed them (riccati_backward_impl, phases C + D): the
// cross-lane form of the B -> C exchange the round-5 review asked to be measured.  0 (shipped): through LDS.  Same bits.
#ifndef TPL_ILQR_GAIN_READLANE
#define TPL_ILQR_GAIN_READLANE 0
#endif

// ---- backward Riccati pass, optim.c:926-985, as a function of its own (NOT inlined) -- same reason as
// line_search_rollouts: inside k_ilqr_update its loop ran with 85 AGPR moves, 30 SGPR-spill reloads
// and 6 scratch stores per stage (of 513 instructions), all issued by the one wavefront of the SIMD.
// The LDS block comes in as an LDS pointer (ds_* addressing in here too), everything uniform is
// made uniform again with readfirstlane (function arguments count as divergent).
typedef __attribute__((address_space(3))) Lds LdsAS;
typedef const __attribute__((address_space(1))) double* GlbCD;    // global-memory pointers: global_load /
typedef __attribute__((address_space(1))) double* GlbD;           // global_store (a generic pointer would be a
typedef __attribute__((address_space(3))) double* LdsD;
typedef const __attribute__((address_space(3))) double* LdsCD;           // flat access, which also counts as an LDS one)
struct RiccatiArgs {
    const double* FX; const double* LXX; const double* FU; const double* LUX; const double* LX; const double* LU;
    const double* LUU; const double* U_; const double* UMAX; const double* UMIN;
    double* KFF; double* KFB; double* KL; double* KBL;
    double mu;
    int T;
    int lt;
    long long* stamp_acc;      // diagnostic builds only: this problem's stamp slots
};

__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ GlbCD uni_g(const double* p) { return (GlbCD)uni_u64((unsigned long long)p); }
__device__ __forceinline__ GlbD uni_g(double* p) { return (GlbD)uni_u64((unsigned long long)p); }
// a generic pointer into LDS -> LDS pointer: the low half of the flat address IS the LDS offset.  (A
// plain address-space cast also emits a null test against the shared aperture, which this hipcc
// mis-selects in some kernels: "Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base".)
__device__ __forceinline__ LdsCD uni_l(const double* p) { return (LdsCD)(unsigned long long)(unsigned)uni_i((int)(unsigned)(unsigned long long)p); }
__device__ __forceinline__ LdsD uni_l(double* p) { return (LdsD)(unsigned long long)(unsigned)uni_i((int)(unsigned)(unsigned long long)p); }
__device__ __forceinline__ double uni_d(double d)
{
    return __longlong_as_double((long long)uni_u64((unsigned long long)__double_as_longlong(d)));
}

__device__ __forceinline__ void riccati_backward_impl(const RiccatiArgs& A, LdsAS* s)
{
    const int lane = (int)threadIdx.x;
    const GlbCD FX = uni_g(A.FX), LXX = uni_g(A.LXX), FU = uni_g(A.FU), LUX = uni_g(A.LUX), LX = uni_g(A.LX),
                LU = uni_g(A.LU), LUU = uni_g(A.LUU), U_ = uni_g(A.U_), UMAX = uni_g(A.UMAX), UMIN = uni_g(A.UMIN);
    const GlbD KFF = uni_g(A.KFF), KFB = uni_g(A.KFB);
    const LdsD KL = uni_l(A.KL), KBL = uni_l(A.KBL);          // the LDS mirrors of the gains (used when lt)
    const double mu = uni_d(A.mu);
    const int T = uni_i(A.T);
    const bool lt = uni_i(A.lt) != 0;
#ifdef TPL_ILQR_STAMP
    long long bacc[5] = {0, 0, 0, 0, 0};
    long long b_last = (long long)__builtin_amdgcn_s_memtime();
#define BSTAMP(i) do { const long long now__ = (long long)__builtin_amdgcn_s_memtime(); bacc[i] += now__ - b_last; b_last = now__; } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif
    // What a stage needs from memory, and where it goes (one stage ahead, so the latency of stage
    // t-1 hides under the dependent phases of stage t):
    //   [fx | fu]  -> registers (lane = entry) -> LDS block F at the end of stage t: every lane reads
    //                 whole columns of it in phases A and B;
    //   the entries of lxx / lux / luu / lx / lu are addends of exactly one Q entry each: the lane
    //                 that owns the Q entry keeps them in registers, they never touch LDS;
    //   u, uMax, uMin: registers of the phase-C lanes.
    // Every lane walks its own pointers down the stages (one 64-bit add each); lanes without an
    // entry load a valid neighbour and park it in a dump slot, so nothing in the loop needs a
    // lane mask of its own.
    constexpr int WS = NXU + 1;                        // row stride of W: column NXU carries Vx
    constexpr int E_XX = (NX * NX + WAVE - 1) / WAVE, E_XU = (NX * NU + WAVE - 1) / WAVE;
    constexpr int N_XX = NX * (NX + 1) / 2, N_UX = NU * NX, N_UU = NU * (NU + 1) / 2;
    constexpr int N_B = N_XX + N_UX + N_UU + NXU;
    constexpr int E_B = (N_B + WAVE - 1) / WAVE, E_A = (NX * NXU + WAVE - 1) / WAVE;
    static_assert(NX * NXU <= 2 * WAVE && N_B <= 2 * WAVE, "state dimension too large for this kernel");
    static_assert(NX * NX + NX <= WAVE || NX > 7, "phase D lane map");
    static_assert(NU * (NX + 1) <= WAVE, "phase C lane map");

    const LdsD dump = &s->dump[0];
    // ---- F: sources and LDS destinations
    GlbCD p_fx[E_XX], p_fu[E_XU];
    LdsD d_fx[E_XX], d_fu[E_XU];
    double r_fx[E_XX], r_fu[E_XU];
#pragma unroll
    for (int q = 0; q < E_XX; ++q) {
        const int e = lane + q * WAVE, ec = e < NX * NX ? e : NX * NX - 1;
        const int i = ec / NX, j = ec - i * NX;
        p_fx[q] = FX + (int64_t)(T - 1) * NX * NX + ec;
        d_fx[q] = e < NX * NX ? &s->F[i * NXU + j] : dump;
    }
#pragma unroll
    for (int q = 0; q < E_XU; ++q) {
        const int e = lane + q * WAVE, ec = e < NX * NU ? e : NX * NU - 1;
        const int i = ec / NU, j = ec - i * NU;
        p_fu[q] = FU + (int64_t)(T - 1) * NX * NU + ec;
        d_fu[q] = e < NX * NU ? &s->F[i * NXU + NX + j] : dump;
    }
    // ---- phase B: entry -> columns of F and W, the two addends, the one or two destinations
    int b_colF[E_B], b_colW[E_B];
    LdsD b_dst0[E_B], b_dst1[E_B];
    GlbCD p_l0[E_B], p_l1[E_B];
    long long b_step[E_B];                             // doubles per stage of the array the addends come from
    double r_l0[E_B], r_l1[E_B];
#pragma unroll
    for (int q = 0; q < E_B; ++q) {
        const int e = lane + q * WAVE;
        int colF = 0, colW = 0;
        GlbCD l0 = LX, l1 = LX;
        long long step = NX;
        LdsD o0 = dump, o1 = dump;
        if (e < N_XX) {
            int hi = 0;
            while ((hi + 1) * (hi + 2) / 2 <= e) ++hi;                 // row of the packed lower triangle
            const int lo = e - hi * (hi + 1) / 2;
            colF = hi; colW = lo; step = NX * NX;
            l0 = LXX + hi * NX + lo; l1 = LXX + lo * NX + hi;
            o0 = &s->Qxx[hi * NX + lo]; o1 = &s->Qxx[lo * NX + hi];
        } else if (e < N_XX + N_UX) {
            const int r = e - N_XX, d = r / NX, jj = r - d * NX;
            colF = NX + d; colW = jj; step = NU * NX;
            l0 = l1 = LUX + r;
            o0 = o1 = &s->Qux[r];
        } else if (e < N_XX + N_UX + N_UU) {
            const int r = e - N_XX - N_UX;
            int hi = 0;
            while ((hi + 1) * (hi + 2) / 2 <= r) ++hi;
            const int lo = r - hi * (hi + 1) / 2;
            colF = NX + hi; colW = NX + lo; step = NU * NU;
            l0 = LUU + hi * NU + lo; l1 = LUU + lo * NU + hi;
            o0 = &s->Quu[hi * NU + lo]; o1 = &s->Quu[lo * NU + hi];
        } else if (e < N_B) {
            const int c = e - (N_XX + N_UX + N_UU);                   // q = l + F^T Vx: Vx is column NXU of W
            colF = c; colW = NXU;
            if (c < NX) { l0 = l1 = LX + c; step = NX; } else { l0 = l1 = LU + (c - NX); step = NU; }
            o0 = o1 = &s->q[c];
        }
        b_colF[q] = colF; b_colW[q] = colW; b_dst0[q] = o0; b_dst1[q] = o1;
        b_step[q] = -step;
        p_l0[q] = l0 + (int64_t)(T - 1) * step; p_l1[q] = l1 + (int64_t)(T - 1) * step;
    }
    // ---- phase C lanes: (d, c), c == NX is the feed-forward entry
    const bool c_lane = lane < NU * (NX + 1);
    const int c_d = c_lane ? lane / (NX + 1) : 0, c_c = c_lane ? lane - c_d * (NX + 1) : 0;
    GlbCD p_u = U_ + (int64_t)(T - 1) * NU, p_umax = UMAX + (int64_t)(T - 1) * NU, p_umin = UMIN + (int64_t)(T - 1) * NU;
    double r_u[NU], r_umax[NU], r_umin[NU];                         // every lane holds the limits of the stage
    const int c_col = (c_c == NX) ? 0 : c_c;                        // column of K a store lane writes
    const int c_step = (c_c == NX) ? NU : NU * NX;                  // doubles per stage of the lane's gain array
    GlbD c_glb = (c_c == NX) ? KFF + (int64_t)(T - 1) * NU + c_d
                             : KFB + (int64_t)(T - 1) * NU * NX + c_d * NX + (c_c == NX ? 0 : c_c);
    LdsD c_mir = (c_c == NX) ? KL + (int64_t)(T - 1) * NU + c_d
                             : KBL + (int64_t)(T - 1) * NU * NX + c_d * NX + (c_c == NX ? 0 : c_c);
    // ---- phase D: lanes [0, NX*NX) own Vxx entries, lanes [NX*NX, NX*NX+NX) own Vx entries
    const bool d_vxx = lane < NX * NX, d_vx = lane >= NX * NX && lane < NX * NX + NX;
    const int d_i = d_vxx ? lane / NX : (d_vx ? lane - NX * NX : 0);
    const int d_j = d_vxx ? lane - (lane / NX) * NX : 0;
    const LdsD d_dst = d_vxx ? &s->Vxx[lane] : (d_vx ? &s->W[d_i * WS + NXU] : dump);
    const LdsCD d_tail = d_vxx ? (LdsCD)&s->Qxx[lane] : (LdsCD)&s->q[d_i];

    auto fetch_F = [&]() {
#pragma unroll
        for (int q = 0; q < E_XX; ++q) { r_fx[q] = *p_fx[q]; p_fx[q] -= NX * NX; }
#pragma unroll
        for (int q = 0; q < E_XU; ++q) { r_fu[q] = *p_fu[q]; p_fu[q] -= NX * NU; }
    };
    auto park_F = [&]() {
#pragma unroll
        for (int q = 0; q < E_XX; ++q) *d_fx[q] = r_fx[q];
#pragma unroll
        for (int q = 0; q < E_XU; ++q) *d_fu[q] = r_fu[q];
    };
    auto fetch_L = [&]() {
#pragma unroll
        for (int q = 0; q < E_B; ++q) {
            r_l0[q] = *p_l0[q]; r_l1[q] = *p_l1[q];
            p_l0[q] += b_step[q]; p_l1[q] += b_step[q];
        }
    };
    auto fetch_ulim = [&]() {
#pragma unroll
        for (int d = 0; d < NU; ++d) { r_u[d] = p_u[d]; r_umax[d] = p_umax[d]; r_umin[d] = p_umin[d]; }
        p_u -= NU; p_umax -= NU; p_umin -= NU;
    };

    if (lane < NX) s->W[lane * WS + NXU] = s->Vx[lane];
    fetch_F();
    park_F();
    fetch_L();
    fetch_ulim();
    lds_sync();
    if (T > 1) fetch_F();
    for (int t = T - 1; t >= 0; --t) {
        BSTAMP(0);
        // phase A: W = Vxx [fx fu]
#pragma unroll
        for (int q = 0; q < E_A; ++q) {
            // (no lane mask: lanes past the last entry repeat it -- same value, same slot)
            const int e = min(lane + q * WAVE, NX * NXU - 1);
            const int i = e / NXU, j = e - i * NXU;
            s->W[i * WS + j] = dot_lds<NX>(&s->Vxx[i * NX], 1, &s->F[j], NXU);
        }
        lds_sync();
        BSTAMP(1);

        // phase B: Q = L + F^T W, q = l + F^T Vx.  Only the lower triangles of the symmetric blocks
        // are formed and then mirrored, like mul_symm (optim.c:159-180).
        double b_val0 = 0.0, b_val1 = 0.0;               // (TPL_ILQR_GAIN_READLANE: the lane's own Q entry and its mirror image)
#pragma unroll
        for (int q = 0; q < E_B; ++q) {
            // (no lane masks: lanes without an entry work on column 0 and write the dump slot; an
            // entry without a mirror image has dst1 == dst0 and l1 == l0: the same value twice)
            const double acc = dot_lds<NX>(&s->F[b_colF[q]], NXU, &s->W[b_colW[q]], WS);
            const double v0 = r_l0[q] + acc, v1 = r_l1[q] + acc;
            *b_dst0[q] = v0;
            *b_dst1[q] = v1;
            if (q == 0) { b_val0 = v0; b_val1 = v1; }
        }
        lds_sync();
        // F of stage t has been read for the last time: [fx | fu] of stage t-1 (in registers since
        // the stage before) takes its place.  Loads and stores of a wavefront complete in order, so
        // this wait sits BEFORE the gain stores of phase C and after nothing recent.
        if (t > 0) park_F();
        BSTAMP(2);

        // phases C + D in one: gains (solve_action optim.c:243-291), control-box projection
        // (:950-963) and the value function update (:965-984, same order of additions).  The gains
        // are a handful of operations on six shared numbers, so EVERY lane forms them -- on a SIMD
        // that costs what sixteen lanes cost -- together with the columns of K it needs; the
        // sixteen (d, c) lanes also store them.  One LDS exchange per stage less.
        {
            double quu[NU * NU], qu[NU], Qi[NU], Qj[NU], Qc[NU];
#if TPL_ILQR_GAIN_READLANE
            // A/B leg (round 6): the six numbers every lane's gain solve shares -- Quu, Qu -- straight from the registers of the
            // phase-B lanes that formed them (v_readlane into scalar registers) instead of through the LDS write -> read of the
            // B -> C exchange; the division chain can start an LDS latency earlier.  Same values, same bits.
            if constexpr (E_B == 1 && N_B <= WAVE) {
                const auto bc = [](double v, int src) {
                    const long long b = __double_as_longlong(v);
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
                    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                };
#pragma unroll
                for (int hi = 0; hi < NU; ++hi)
#pragma unroll
                    for (int lo = 0; lo <= hi; ++lo) {
                        const int src = N_XX + N_UX + hi * (hi + 1) / 2 + lo;
                        quu[hi * NU + lo] = bc(b_val0, src);
                        quu[lo * NU + hi] = bc(b_val1, src);
                    }
#pragma unroll
                for (int m = 0; m < NU; ++m) qu[m] = bc(b_val0, N_XX + N_UX + N_UU + NX + m);
            } else
#endif
            {
#pragma unroll
                for (int m = 0; m < NU * NU; ++m) quu[m] = s->Quu[m];
#pragma unroll
                for (int m = 0; m < NU; ++m) qu[m] = s->q[NX + m];
            }
#pragma unroll
            for (int m = 0; m < NU; ++m) {
                Qi[m] = s->Qux[m * NX + d_i]; Qj[m] = s->Qux[m * NX + d_j]; Qc[m] = s->Qux[m * NX + c_col];
            }
            const double tail = *d_tail;
            __builtin_amdgcn_sched_barrier(0);
            double inv[NU][NU];
            if constexpr (NU == 1) {
                inv[0][0] = (quu[0] > 0.0) ? -1.0 / (quu[0] + mu) : 0.0;
            } else {
                const double qa = quu[0] + mu, qb = quu[1], qd = quu[NU + 1] + mu;
                const double det = qa * qd - qb * qb;
                const double inv_det = -1.0f / det;
                inv[0][0] = qd * inv_det;
                inv[0][NU - 1] = -qb * inv_det;
                inv[NU - 1][0] = inv[0][NU - 1];
                inv[NU - 1][NU - 1] = qa * inv_det;
            }
            double kk[NU], Ki[NU], Kj[NU], Kc[NU];
#pragma unroll
            for (int d = 0; d < NU; ++d) {
                double kd = 0.0, ki = 0.0, kj = 0.0, kc = 0.0;
#pragma unroll
                for (int m = 0; m < NU; ++m) {
                    kd = fma(inv[d][m], qu[m], kd);
                    ki = fma(inv[d][m], Qi[m], ki);
                    kj = fma(inv[d][m], Qj[m], kj);
                    kc = fma(inv[d][m], Qc[m], kc);
                }
                const double cd = r_u[d] + kd;
                const bool hi = cd > r_umax[d], lo = cd < r_umin[d];
                kk[d] = lo ? r_umin[d] - r_u[d] : (hi ? r_umax[d] - r_u[d] : kd);
                Ki[d] = (hi || lo) ? 0.0 : ki;
                Kj[d] = (hi || lo) ? 0.0 : kj;
                Kc[d] = (hi || lo) ? 0.0 : kc;
            }
            if (c_lane) {
                // one store path for feed-forward and feedback entries: the lane walks its own pointers
                double val = (c_c == NX) ? kk[0] : Kc[0];
#pragma unroll
                for (int d = 1; d < NU; ++d) val = (c_d == d) ? ((c_c == NX) ? kk[d] : Kc[d]) : val;
                *c_glb = val;
                if (lt) *c_mir = val;
                c_glb -= c_step;
                c_mir -= c_step;
            }
            // ALL requests of the stage in one place, after the gain stores: stage t-2 of F, stage
            // t-1 of the addends and limits.  Everything the loop waits for was then issued most of
            // a stage ago.
            if (t > 1) fetch_F();
            if (t > 0) { fetch_L(); fetch_ulim(); }

            double kq[NU];                                   // (K^T Quu)[i][d]
#pragma unroll
            for (int d = 0; d < NU; ++d) {
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < NU; ++m) a = fma(Ki[m], quu[m * NU + d], a);
                kq[d] = a;
            }
            double mij = 0.0, mji = 0.0;
#pragma unroll
            for (int d = 0; d < NU; ++d) { mij = fma(Ki[d], Qj[d], mij); mji = fma(Kj[d], Qi[d], mji); }
            double v1 = mji + mij;
#pragma unroll
            for (int d = 0; d < NU; ++d) v1 = fma(kq[d], Kj[d], v1);
            double v2 = 0.0;
#pragma unroll
            for (int d = 0; d < NU; ++d) v2 = fma(kq[d], kk[d], v2);
#pragma unroll
            for (int d = 0; d < NU; ++d) v2 = fma(Ki[d], qu[d], v2);
#pragma unroll
            for (int d = 0; d < NU; ++d) v2 = fma(Qi[d], kk[d], v2);
            *d_dst = (d_vxx ? v1 : v2) + tail;               // Vxx entry, or Vx where phase B reads it
        }
        lds_sync();
        BSTAMP(4);
    }
#ifdef TPL_ILQR_STAMP
    if (lane == 0 && A.stamp_acc) for (int i = 0; i < 5; ++i) A.stamp_acc[6 + i] += bacc[i];
#endif
}

// the call boundary k_ilqr_update needs (see above); k_split_chain, which holds nothing else, takes the body inline
__device__ __attribute__((noinline)) void riccati_backward(const RiccatiArgs A, LdsAS* s) { riccati_backward_impl(A, s); }

// ---- the same pass for problems with two or three states and one input (tpl's lateral and speed profile problems: what
// its default planner solves, one problem per call, twice per cycle).  With matrices this small the lane-parallel form
// above is all exchange and no work: three LDS round trips per stage for a dozen multiplications.  Here every lane
// carries the whole recursion in registers -- Vxx, Vx, the stage's derivatives one stage ahead -- and a stage is one
// straight dependent chain without a wait in it; lane 0 stores the gains.  Same operations in the same order as the
// lane-parallel form (the dots accumulate from zero in index order, only the lower triangle of Qxx is formed and
// mirrored), so both give the same bits.
#ifndef TPL_ILQR_SMALL_NX
#define TPL_ILQR_SMALL_NX 2
#endif
constexpr bool SMALL_BACKWARD = (NU == 1 && NX <= TPL_ILQR_SMALL_NX);

struct SmallStage {
    double fx[NX][NX], fu[NX], lxx[NX][NX], lux[NX], luu, lx[NX], lu, u, umax, umin;
};

__device__ __attribute__((noinline)) void riccati_backward_small(const RiccatiArgs A, LdsAS* s)
{
    const GlbCD FX = uni_g(A.FX), LXX = uni_g(A.LXX), FU = uni_g(A.FU), LUX = uni_g(A.LUX), LX = uni_g(A.LX),
                LU = uni_g(A.LU), LUU = uni_g(A.LUU), U_ = uni_g(A.U_), UMAX = uni_g(A.UMAX), UMIN = uni_g(A.UMIN);
    const GlbD KFF = uni_g(A.KFF), KFB = uni_g(A.KFB);
    const LdsD KL = uni_l(A.KL), KBL = uni_l(A.KBL);
    const double mu = uni_d(A.mu);
    const int T = uni_i(A.T);
    const bool lt = uni_i(A.lt) != 0;
    const bool writer = threadIdx.x == 0;

    double Vxx[NX][NX], Vx[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        Vx[i] = s->Vx[i];
#pragma unroll
        for (int j = 0; j < NX; ++j) Vxx[i][j] = s->Vxx[i * NX + j];
    }
    // Addresses: a lone wavefront pays four cycles for a scalar add as for a vector one, and ten arrays walked by
    // 64-bit scalar pointers cost two dozen of them per stage.  Three byte offsets in vector registers (one per row
    // width), added to the arrays' scalar bases by the load instruction itself, cost three.
    typedef const __attribute__((address_space(1))) char* GlbCC;
    unsigned lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));                      // a vector register the compiler cannot fold away
    auto at = [&](GlbCD base, unsigned byte_off, int k) { return *(GlbCD)((GlbCC)base + byte_off + 8u * (unsigned)k); };
    auto fetch = [&](int t, SmallStage& g) {
        const unsigned o1 = lane_zero + (unsigned)t * 8u, o2 = lane_zero + (unsigned)t * (8u * NX),
                       o4 = lane_zero + (unsigned)t * (8u * NX * NX);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            g.fu[i] = at(FU, o2, i); g.lux[i] = at(LUX, o2, i); g.lx[i] = at(LX, o2, i);
#pragma unroll
            for (int j = 0; j < NX; ++j) { g.fx[i][j] = at(FX, o4, i * NX + j); g.lxx[i][j] = at(LXX, o4, i * NX + j); }
        }
        g.luu = at(LUU, o1, 0); g.lu = at(LU, o1, 0); g.u = at(U_, o1, 0); g.umax = at(UMAX, o1, 0); g.umin = at(UMIN, o1, 0);
    };
    // One stage of the recursion on the derivatives in `cur`.
    auto stage = [&](int t, const SmallStage& cur) {
        auto F = [&](int m, int j) { return j < NX ? cur.fx[m][j < NX ? j : 0] : cur.fu[m]; };
        // W = Vxx [fx fu]
        double W[NX][NXU];
#pragma unroll
        for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NXU; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < NX; ++m) acc = fma(Vxx[i][m], F(m, j), acc);
                W[i][j] = acc;
            }
        // Q = L + F^T W (lower triangle of Qxx, mirrored), q = l + F^T Vx
        double Qxx[NX][NX], Qux[NX], qx[NX];
#pragma unroll
        for (int hi = 0; hi < NX; ++hi)
#pragma unroll
            for (int lo = 0; lo <= hi; ++lo) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < NX; ++m) acc = fma(cur.fx[m][hi], W[m][lo], acc);
                Qxx[hi][lo] = cur.lxx[hi][lo] + acc;
                Qxx[lo][hi] = cur.lxx[lo][hi] + acc;
            }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) acc = fma(cur.fu[m], W[m][j], acc);
            Qux[j] = cur.lux[j] + acc;
        }
        double quu, qu;
        {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) acc = fma(cur.fu[m], W[m][NX], acc);
            quu = cur.luu + acc;
        }
#pragma unroll
        for (int c = 0; c < NX; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) acc = fma(cur.fx[m][c], Vx[m], acc);
            qx[c] = cur.lx[c] + acc;
        }
        {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) acc = fma(cur.fu[m], Vx[m], acc);
            qu = cur.lu + acc;
        }
        // gains (solve_action optim.c:243-291) and the control-box projection (:950-963)
        // (selects, no branches: the stage stays one basic block, so the loads of later stages can be scheduled across it)
        double quotient = -1.0 / (quu + mu), room_up = cur.umax - cur.u, room_down = cur.umin - cur.u;
        asm volatile("" : "+v"(quotient), "+v"(room_up), "+v"(room_down));
        const double inv = (quu > 0.0) ? quotient : 0.0;
        const double kd = fma(inv, qu, 0.0);
        double K[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) K[c] = fma(inv, Qux[c], 0.0);
        const double cd = cur.u + kd;
        const bool over = cd > cur.umax, under = cd < cur.umin;
        const double kk = under ? room_down : (over ? room_up : kd);
#pragma unroll
        for (int c = 0; c < NX; ++c) K[c] = (over || under) ? 0.0 : K[c];
        if (writer) {
            KFF[t] = kk;
#pragma unroll
            for (int c = 0; c < NX; ++c) KFB[(int64_t)t * NX + c] = K[c];
            if (lt) {
                KL[t] = kk;
#pragma unroll
                for (int c = 0; c < NX; ++c) KBL[t * NX + c] = K[c];
            }
        }
        // value function (:965-984, same order of additions)
        double nVxx[NX][NX], nVx[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const double kq = fma(K[i], quu, 0.0);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const double mij = fma(K[i], Qux[j], 0.0), mji = fma(K[j], Qux[i], 0.0);
                double v1 = mji + mij;
                v1 = fma(kq, K[j], v1);
                nVxx[i][j] = v1 + Qxx[i][j];
            }
            double v2 = fma(kq, kk, 0.0);
            v2 = fma(K[i], qu, v2);
            v2 = fma(Qux[i], kk, v2);
            nVx[i] = v2 + qx[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            Vx[i] = nVx[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) Vxx[i][j] = nVxx[i][j];
        }
    };
    // The derivatives come from global memory (L2: several hundred cycles), a stage's chain is shorter than that: four
    // stage records are kept in flight, each refilled for stage t-4 as soon as stage t has consumed it.  The main loop
    // is one basic block (four stages, unconditional refills with the stage index clamped at 0), so the compiler
    // counts the outstanding loads exactly; the T % 4 stages before it run one at a time.
    SmallStage b0, b1, b2, b3;
    int t = T - 1;
    for (; t >= 0 && ((t + 1) & 3) != 0; --t) {
        fetch(t, b0);
        stage(t, b0);
    }
    if (t >= 3) {
        fetch(t, b0); fetch(t - 1, b1); fetch(t - 2, b2); fetch(t - 3, b3);
        for (; t >= 3; t -= 4) {
            stage(t, b0);
            fetch(max(t - 4, 0), b0);
            stage(t - 1, b1);
            fetch(max(t - 5, 0), b1);
            stage(t - 2, b2);
            fetch(max(t - 6, 0), b2);
            stage(t - 3, b3);
            fetch(max(t - 7, 0), b3);
        }
    }
    lds_sync();
}

// The parameter record through the constant address space, field by field (typed, so the copy
// ends up in registers and not in a stack slot).
__device__ __forceinline__ void load_params_uniform(Params& dst, const Params* src_generic)
{
    typedef const __attribute__((address_space(4))) char* CstC;
    typedef const __attribute__((address_space(4))) double* CstD;
    typedef const __attribute__((address_space(4))) long long* CstL;
    typedef const __attribute__((address_space(4))) int* CstI;
    const CstC base = (CstC)uni_u64((unsigned long long)src_generic);
#define TPL_PARAM_S(n) dst.n = *(CstD)(base + offsetof(Params, n));
#define TPL_PARAM_A(n) { const CstC q = base + offsetof(Params, n); \
        dst.n.data = (const double*)(unsigned long long)*(CstL)(q + offsetof(DynArr, data)); \
        dst.n.n0 = *(CstL)(q + offsetof(DynArr, n0)); dst.n.n1 = *(CstL)(q + offsetof(DynArr, n1)); \
        dst.n.ndim = *(CstI)(q + offsetof(DynArr, ndim)); dst.n.pad = 0; }
    TPL_PARAM_LIST
#undef TPL_PARAM_S
#undef TPL_PARAM_A
}

// The line-search rollouts as a function of their own (NOT inlined): inside k_ilqr_update the
// derivative refresh drives the register allocation to 500 registers with most long-lived values
// parked in AGPRs, and the rollout loop -- a chain of dependent instructions on a single wavefront,
// where every extra move is paid in full -- would shuttle its operands through them.  A real call
// gives the loop a register file of its own; the call itself happens once per line search.
struct RolloutArgs {
    double* CX; double* CU;
    const double* X_; const double* KFFr; const double* Ur; const double* UMAXr; const double* UMINr;
    const double* KFBr; const double* Xr;
    const Params* P;
    double dt;
    int rows, opt_start, T, int_type;
    bool second_order;
    int direct;                // DevState::roll_direct
    long long* stamp_acc;      // diagnostic builds only
};

template <bool LT, bool SECOND_ORDER, int INT_TYPE>
__device__ __forceinline__ void line_search_rollouts_impl(const RolloutArgs& A, LdsAS* s)
{
    // Candidate c runs on the QUAD of lanes 4c .. 4c+3: all four hold the same state and
    // control, the model's dynamics spread their independent look-ups and sin/cos evaluations
    // over them (dynamics_quad); all four record the candidate (same values, same LDS slots --
    // a lane mask around the ring writes costs more than the redundant writes).
    //
    // Where the results go matters more than it looks: loads and stores of a wavefront share ONE
    // counter (vmcnt) and complete in order, so a load that follows a store to global memory
    // waits for that store's acknowledgement from L2 -- with nine stores per stage every stage
    // paid a full write round trip (0.5 ms of a 5 ms update).  The stage results therefore
    // collect in an LDS ring and leave every ROLL_RING stages in one coalesced sweep of all 64
    // lanes; the operands of the feedback law are read through pointers of the address space they
    // live in (LDS mirrors when LT, else global memory), not through flat accesses.
    using RP = typename std::conditional<LT, LdsCD, GlbCD>::type;
    const int lane = (int)threadIdx.x;
    const GlbD CX = uni_g(A.CX), CU = uni_g(A.CU);
    const GlbCD X_ = uni_g(A.X_);
    RP KFFr, Ur, UMAXr, UMINr, KFBr, Xr;
    if constexpr (LT) {
        KFFr = uni_l(A.KFFr); Ur = uni_l(A.Ur); UMAXr = uni_l(A.UMAXr); UMINr = uni_l(A.UMINr);
        KFBr = uni_l(A.KFBr); Xr = uni_l(A.Xr);
    } else {
        KFFr = uni_g(A.KFFr); Ur = uni_g(A.Ur); UMAXr = uni_g(A.UMAXr); UMINr = uni_g(A.UMINr);
        KFBr = uni_g(A.KFBr); Xr = uni_g(A.Xr);
    }
    // The parameter record is read-only for the whole launch: fetch it ONCE, with scalar loads from
    // the constant address space, into registers.  Through the generic pointer every use inside the
    // dynamics was a flat load on the dependent chain (nothing can be hoisted over the ring stores).
    Params Pl;
    load_params_uniform(Pl, A.P);
    const Params* P = &Pl;
    const double dt = uni_d(A.dt);
    const int rows = uni_i(A.rows), opt_start = uni_i(A.opt_start), T = uni_i(A.T);
    constexpr int int_type = INT_TYPE;                 // compile time: the integrator's branches fold
    constexpr bool second_order = SECOND_ORDER;        // compile time: two uniform branches per stage less

#ifdef TPL_ILQR_STAMP
    long long racc[3] = {0, 0, 0};
    long long r_last = (long long)__builtin_amdgcn_s_memtime();
#define RSTAMP(i) do { const long long now__ = (long long)__builtin_amdgcn_s_memtime(); racc[i] += now__ - r_last; r_last = now__; } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif
    const bool active = lane < 4 * N_ALPHA;
    const int cand = lane >> 2;
    // Where a stage's results go.  DIRECT (round 6): the loop reads nothing from global memory, so its stores block nothing --
    // every quad writes its candidate's (x_{t+1}, u_t) straight into the candidate arrays (all four lanes: same value, same
    // address), nine stores and two pointer bumps per stage instead of nine ring writes plus a flush of 576 doubles every
    // eight stages behind two fences.  Else through the ring (the reason it exists: see above).
    const bool direct = LT && uni_i(A.direct) != 0;
    GlbD px = CX + ((int64_t)(active ? cand : 0) * rows + opt_start + 1) * NX;
    GlbD pu = CU + ((int64_t)(active ? cand : 0) * rows + opt_start) * NU;
    double a = 1.0;
    for (int i = 0; i < cand; ++i) a *= 10.0;
    a = 1.0 / a;                                       // 1.0 / pow(10, i)
    double x[NX], xn[NX], u[NU];
    // stages up to opt_start are frozen (tpl never moves opt_start off 0)
    for (int e = lane; e < N_ALPHA * (opt_start + 1) * NX; e += WAVE) {
        const int c = e / ((opt_start + 1) * NX), off = e - c * ((opt_start + 1) * NX);
        CX[(int64_t)c * rows * NX + off] = X_[off];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)opt_start * NX + i];

#if TPL_ILQR_ABLATE == 4
    const int T_roll = opt_start + (T - opt_start) / 2;
#else
    const int T_roll = T;
#endif
#if TPL_ILQR_ABLATE == 5
    for (int pass = 0; pass < 2; ++pass) {
    const int T_pass = pass == 0 ? T_roll : opt_start + (T_roll - opt_start) / 2;
    if (pass == 1) {
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)opt_start * NX + i];
    }
#else
    const int T_pass = T_roll;
#endif
    for (int t0 = opt_start; t0 < T_pass; t0 += ROLL_RING) {
        const int n = (T_pass - t0 < ROLL_RING) ? T_pass - t0 : ROLL_RING;
        if (active) {
            for (int k = 0; k < n; ++k) {
                const int t = t0 + k;
                // feedback law: lane q of the quad forms control q % NU (its own row of K, read through
                // a lane-dependent address), the quad then exchanges the controls with DPP moves --
                // half the arithmetic and loads of every lane forming both.  All operands first, then
                // the arithmetic (one memory round trip); the order of the additions is the reference's.
                {
                    const int my_d = (NU > 1 && NU <= 4) ? (lane & 3) % NU : 0;
                    const double kff = KFFr[(int64_t)t * NU + my_d];
                    const double ub = Ur[(int64_t)t * NU + my_d];
                    const double hi = UMAXr[(int64_t)t * NU + my_d];
                    const double lo = UMINr[(int64_t)t * NU + my_d];
                    double kfb[NX], xr[NX];
                    if (second_order) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) kfb[i] = KFBr[(int64_t)t * NU * NX + my_d * NX + i];
#pragma unroll
                        for (int i = 0; i < NX; ++i) xr[i] = Xr[(int64_t)t * NX + i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double ud;
                    if (second_order) {
                        ud = kff * a;
                        ud = ud + ub;
#pragma unroll
                        for (int i = 0; i < NX; ++i) ud += kfb[i] * (x[i] - xr[i]);
                        ud = hi < ud ? hi : ud;
                        ud = lo > ud ? lo : ud;
                    } else {
                        ud = ub - kff * a;
                    }
#pragma unroll
                    for (int d = 0; d < NU; ++d) u[d] = (NU > 1 && NU <= 4) ? tpl_quad_bcast(ud, d) : ud;
                    if (direct) {
#pragma unroll
                        for (int d = 0; d < NU; ++d) pu[d] = u[d];
                        pu += NU;
                    } else {
#pragma unroll
                        for (int d = 0; d < NU; ++d) s->ring_u[ring_ix(cand, k, d, NU)] = u[d];    // (all four lanes: same value, same slot)
                    }
                }
                RSTAMP(0);
                dynamics_quad(*P, x, u, t, dt, int_type, xn);
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = xn[i];
                if (direct) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) px[i] = xn[i];
                    px += NX;
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i) s->ring_x[ring_ix(cand, k, i, NX)] = xn[i];
                }
                RSTAMP(1);
            }
        }
        if (direct) continue;
        lds_sync();
        // the ring goes out: candidate c, stages t0 .. t0+n-1 are n*NX (n*NU) consecutive doubles in global memory
        const auto flush = [&](auto n_entry_c, GlbD dst, int64_t t_first, const LdsAS* sp, bool is_x, int e) {
            constexpr int NE = decltype(n_entry_c)::value;
            int c, off, slot;
            if (TPL_ILQR_RING_T) { c = e & (N_ALPHA - 1); off = e / N_ALPHA; slot = e; }
            else { c = e / (n * NE); off = e - c * (n * NE); slot = c * ROLL_RING * NE + off; }
            dst[(int64_t)c * rows * NE + t_first * NE + off] = is_x ? sp->ring_x[slot] : sp->ring_u[slot];
        };
        static_assert((N_ALPHA & (N_ALPHA - 1)) == 0, "candidate index by mask");
        if (n == ROLL_RING) {
#pragma unroll
            for (int q = 0; q < (N_ALPHA * ROLL_RING * NX + WAVE - 1) / WAVE; ++q) {
                const int e = lane + q * WAVE;
                if (e < N_ALPHA * ROLL_RING * NX) flush(std::integral_constant<int, NX>{}, CX, (int64_t)(t0 + 1), s, true, e);
            }
#pragma unroll
            for (int q = 0; q < (N_ALPHA * ROLL_RING * NU + WAVE - 1) / WAVE; ++q) {
                const int e = lane + q * WAVE;
                if (e < N_ALPHA * ROLL_RING * NU) flush(std::integral_constant<int, NU>{}, CU, (int64_t)t0, s, false, e);
            }
        } else {
            for (int e = lane; e < N_ALPHA * n * NX; e += WAVE) flush(std::integral_constant<int, NX>{}, CX, (int64_t)(t0 + 1), s, true, e);
            for (int e = lane; e < N_ALPHA * n * NU; e += WAVE) flush(std::integral_constant<int, NU>{}, CU, (int64_t)t0, s, false, e);
        }
        lds_sync();
        RSTAMP(2);
    }
#if TPL_ILQR_ABLATE == 5
    }
#endif
#ifdef TPL_ILQR_STAMP
    if (lane == 0 && A.stamp_acc) { A.stamp_acc[14] += racc[0]; A.stamp_acc[15] += racc[1]; A.stamp_acc[4] += racc[2]; }
#endif
}

template <bool LT, bool SECOND_ORDER, int INT_TYPE>
__device__ __attribute__((noinline)) void line_search_rollouts(const RolloutArgs A, LdsAS* s)
{
    line_search_rollouts_impl<LT, SECOND_ORDER, INT_TYPE>(A, s);
}

// ---------------------------------------------------------------------------
// The tail of one iteration, shared by k_ilqr_update and the chain phase of the split / fused forms: stage costs of the
// candidates the rollouts left in CX / CU, the accept-first test, and the copies the reference makes after it
// (optim.c:840-873, 960-985).  Returns whether a step was accepted; J, alpha and the two flags are updated.
//   MIRRORS: the x / u mirrors in LDS outlive this call (single launch, fused form) -- keep them current on accept.
//   OLD_X_FROM_MIRROR: read the trajectory being replaced from its mirror (it is current, and nearer).
//   Stamp: diagnostic builds time the cost pass (12) and the ordered sums (13); a no-op otherwise.
// ---------------------------------------------------------------------------
struct IterTail {
    double *X_, *U_, *IS, *NIS, *NXT_X, *NXT_U, *PRV_X, *PRV_K;
    const double *KFF, *LAM, *BW;
    double *CX, *CU, *CC, *XL, *UL;
    int64_t rows;
    int T, opt_start;
    double dt;
    bool lt;
};

template <bool MIRRORS, bool OLD_X_FROM_MIRROR, class Stamp>
__device__ __forceinline__ bool iteration_tail(const IterTail& a, const Params& P, Lds& s, double& J, double& alpha,
                                               bool& traj_changed, bool& improved, Stamp stamp)
{
    const int lane = threadIdx.x;
    const int T = a.T, opt_start = a.opt_start;
    const double dt = a.dt;
    const bool lt = a.lt;
    double* const X_ = a.X_; double* const U_ = a.U_; double* const IS = a.IS; double* const NIS = a.NIS;
    double* const NXT_X = a.NXT_X; double* const NXT_U = a.NXT_U; double* const PRV_X = a.PRV_X; double* const PRV_K = a.PRV_K;
    const double* const KFF = a.KFF; const double* const LAM = a.LAM; const double* const BW = a.BW;
    double* const CX = a.CX; double* const CU = a.CU; double* const CC = a.CC; double* const XL = a.XL; double* const UL = a.UL;
    double Jc = INFINITY;
    // pass B: stage costs and their ordered sums, 64 (candidate, stage) pairs at a time -- in the order the
    // reference tries the step sizes (optim.c:859-873): alpha = 1 first, and the seven smaller steps only when
    // that one does not improve the cost.  Most iterations accept alpha = 1: one pass over the stages instead of
    // seven.  The costs of candidates nobody asks for are never formed; what is formed is formed as before.
    const int cc_rows = T + 1;
    const bool cc_lds = N_ALPHA * cc_rows <= N_ALPHA * ROLL_RING * NX;
    for (int part = 0; part < 2; ++part) {
        const int n_t = T - opt_start;
        const int c_lo = part ? 1 : 0, n_c = part ? N_ALPHA - 1 : 1;
        for (int idx = lane; idx < n_c * n_t; idx += WAVE) {
            const int cand = part ? 1 + idx % (N_ALPHA - 1) : 0;
            const int t = opt_start + (part ? idx / (N_ALPHA - 1) : idx);
            const double* cx = CX + ((int64_t)cand * a.rows + t) * NX;
            const double* cu = CU + ((int64_t)cand * a.rows + t) * NU;
            double x[NX], u[NU], c;
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = cx[i];
#pragma unroll
            for (int d = 0; d < NU; ++d) u[d] = cu[d];
#if TPL_ILQR_ABLATE == 3
            c = x[0] * 1e-9;
#else
            costs(P, x, u, LAM + (int64_t)t * NCM, BW, (double)t, dt, &c, 1);
#if TPL_ILQR_ABLATE == 6
            {
                double c2;
                asm volatile("" : "+v"(x[0]), "+v"(u[0]));       // (a second evaluation the compiler cannot merge with the first)
                costs(P, x, u, LAM + (int64_t)t * NCM, BW, (double)t, dt, &c2, 1);
                asm volatile("" :: "v"(c2));
            }
#endif
#endif
            // into LDS where the rollouts' ring (idle now) can hold them: a store to global
            // memory would put a write acknowledgement in front of the next pass' loads
            if (cc_lds) s.ring_x[cand * cc_rows + t] = c;
            else CC[(int64_t)cand * a.rows + t] = c;
        }
        __syncthreads();
        stamp(12);
        // ordered sums, cost += c for t = optStart..T-1, then the end costs (:776-790)
        if (lane >= c_lo && lane < c_lo + n_c) {
            const double* cc = cc_lds ? &s.ring_x[lane * cc_rows] : CC + (int64_t)lane * a.rows;
            double cost = 0.0, c;
            int t = opt_start;
            for (; t + 8 <= T; t += 8) {                    // eight loads in flight, additions in order
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = cc[t + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) cost += v[q];
            }
            for (; t < T; ++t) cost += cc[t];
            double x[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = CX[((int64_t)lane * a.rows + T) * NX + i];
            endCosts(P, x, (double)T, dt, &c, 1);
            cost += c;
            Jc = cost;
        }
        if (part == 0 && __ballot(lane == 0 && Jc < J && isfinite(Jc) && Jc >= 0.0) != 0ull) break;
    }
    stamp(13);
    // testImprovement, optim.c:840-857
    const bool ok = lane < N_ALPHA && Jc < J && isfinite(Jc) && Jc >= 0.0;
    const unsigned long long okm = __ballot(ok);
    const bool accepted = okm != 0ull;
    const int pick = accepted ? (int)__builtin_ctzll(okm) : N_ALPHA - 1;
    {
        double p10 = 1.0;
        for (int i = 0; i < pick; ++i) p10 *= 10.0;
        alpha = 1.0 / p10;
    }
    __syncthreads();                                   // candidates visible to the whole wave
    const double* cx = CX + (int64_t)pick * a.rows * NX;
    const double* cu = CU + (int64_t)pick * a.rows * NU;
    if (accepted) {
        J = __shfl(Jc, pick, WAVE);
        // (all loads of a block of eight rows before its stores: loads and stores of a
        // wavefront complete in order, a load behind a store waits for the write)
        const double* const old_x = (OLD_X_FROM_MIRROR && lt) ? XL : X_;
        for (int e0 = 0; e0 < (T + 1) * NX; e0 += WAVE * 8) {
            double ox[8], nv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < (T + 1) * NX) { ox[k] = old_x[e]; nv[k] = cx[e]; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < (T + 1) * NX) {
                    PRV_X[e] = ox[k];
                    NXT_X[e] = nv[k];
                    X_[e] = nv[k];
                    if constexpr (MIRRORS) { if (lt) XL[e] = nv[k]; }
                }
            }
        }
        for (int e0 = 0; e0 < T * NU; e0 += WAVE * 4) {
            double ok_[4], nu_[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < T * NU) { ok_[k] = KFF[e]; nu_[k] = (e >= opt_start * NU) ? cu[e] : 0.0; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < T * NU) {
                    PRV_K[e] = ok_[k];
                    if (e >= opt_start * NU) {
                        NXT_U[e] = nu_[k];
                        U_[e] = nu_[k];
                        if constexpr (MIRRORS) { if (lt) UL[e] = nu_[k]; }
                    }
                }
            }
        }
        for (int e = lane; e < T + 1; e += WAVE) { NIS[e] = dt; IS[e] = dt; }
        traj_changed = true;
        improved = true;
    } else {
        // the reference leaves the last tried rollout (alpha = 1e-7) in next_*
        for (int e0 = 0; e0 < (T + 1) * NX; e0 += WAVE * 8) {
            double nv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < (T + 1) * NX) nv[k] = cx[e];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < (T + 1) * NX) NXT_X[e] = nv[k];
            }
        }
        for (int e0 = opt_start * NU; e0 < T * NU; e0 += WAVE * 4) {
            double nu_[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < T * NU) nu_[k] = cu[e];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * WAVE + lane;
                if (e < T * NU) NXT_U[e] = nu_[k];
            }
        }
        for (int e = lane; e < T + 1; e += WAVE) NIS[e] = dt;
    }
    return accepted;
}

// ---------------------------------------------------------------------------
// Three more pieces both kernels run as they stand (k_ilqr_update; the begin / chain phases of the split and fused forms).
// ---------------------------------------------------------------------------
// Initial trajectory rollout and its cost, optim.c:1101-1111.  The dynamics chain is serial in t (one quad, see
// dynamics_quad); the stage costs only need (x_t, u_t), so they are evaluated for all stages in parallel afterwards and
// summed in the reference's order t = optStart..T-1.  Returns J on lane 0 (other lanes: 0).
__device__ __forceinline__ double initial_rollout_cost(const Params& P, double* X_, const double* U_, double* IS, const double* LAM,
                                                       const double* BW, double* CC, int T, int opt_start, double dt, int int_type)
{
    const int lane = threadIdx.x;
    double J = 0.0;
    if (lane < 4) {                                    // one quad: see dynamics_quad
        double x[NX], u[NU], xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)opt_start * NX + i];
        for (int t = opt_start; t < T; ++t) {
#pragma unroll
            for (int d = 0; d < NU; ++d) u[d] = U_[(int64_t)t * NU + d];
            if (lane == 0) IS[t] = dt;
            dynamics_quad(P, x, u, t, dt, int_type, xn);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                x[i] = xn[i];
                if (lane == 0) X_[(int64_t)(t + 1) * NX + i] = xn[i];
            }
        }
        if (lane == 0) IS[T] = dt;
    }
    __syncthreads();
    for (int t = opt_start + lane; t < T; t += WAVE) {
        double x[NX], u[NU], c;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)t * NX + i];
#pragma unroll
        for (int d = 0; d < NU; ++d) u[d] = U_[(int64_t)t * NU + d];
        costs(P, x, u, LAM + (int64_t)t * NCM, BW, (double)t, dt, &c, 1);
        CC[t] = c;
    }
    __syncthreads();
    if (lane == 0) {
        double x[NX], c;
        int t = opt_start;
        for (; t + 8 <= T; t += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = CC[t + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) J += v[q];
        }
        for (; t < T; ++t) J += CC[t];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)T * NX + i];
        endCosts(P, x, (double)T, dt, &c, 1);
        J += c;
    }
    return J;
}

// Multiplier update, optim.c:1119-1132 (lane = stage).
__device__ __forceinline__ void multiplier_update(const Params& P, const double* X_, const double* U_, const double* IS, double* LAM,
                                                  const double* BW, const double* LGL, int T, int opt_start)
{
    const int lane = threadIdx.x;
    if (NC > 0) {
        for (int t = opt_start + lane; t < T; t += WAVE) {
            double x[NX], u[NU], lam[NCM], g[NCM];
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)t * NX + i];
#pragma unroll
            for (int d = 0; d < NU; ++d) u[d] = U_[(int64_t)t * NU + d];
#pragma unroll
            for (int c = 0; c < NCM; ++c) { lam[c] = LAM[(int64_t)t * NCM + c]; g[c] = 0.0; }
            constraints(P, x, u, lam, BW, (double)t, IS[t], g, 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double v = lam[c] + BW[c] * g[c];
                v = v > 0.0 ? v : 0.0;
                v = LGL[c] < v ? LGL[c] : v;
                LAM[(int64_t)t * NCM + c] = v;
            }
        }
    }
}

// First-order backward pass, optim.c:1044-1073 (s.Vx holds the terminal gradient on entry).
__device__ __forceinline__ void first_order_backward(Lds& s, const double* FX, const double* FU, const double* LX, const double* LU,
                                                     const double* U_, const double* UMAX, const double* UMIN, double* G_, double* KFF,
                                                     double* KL, bool lt, int T)
{
    const int lane = threadIdx.x;
    for (int t = T - 1; t >= 0; --t) {
        double qx = 0.0, qu = 0.0;
        if (lane < NX) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) acc += FX[(int64_t)t * NX * NX + m * NX + lane] * s.Vx[m];
            qx = LX[(int64_t)t * NX + lane] + acc;
        }
        if (lane < NU) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) acc += FU[(int64_t)t * NX * NU + m * NU + lane] * s.Vx[m];
            qu = LU[(int64_t)t * NU + lane] + acc;
            G_[(int64_t)t * NU + lane] = qu;
            double kk = qu;
            const double c = U_[(int64_t)t * NU + lane] - qu;
            if (c > UMAX[(int64_t)t * NU + lane]) kk = U_[(int64_t)t * NU + lane] - UMAX[(int64_t)t * NU + lane];
            if (c < UMIN[(int64_t)t * NU + lane]) kk = U_[(int64_t)t * NU + lane] - UMIN[(int64_t)t * NU + lane];
            KFF[(int64_t)t * NU + lane] = kk;
            if (lt) KL[(int64_t)t * NU + lane] = kk;
        }
        __syncthreads();
        if (lane < NX) s.Vx[lane] = qx;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// one update() of problem blockIdx.x  (optim.c:1091-1160)
// ---------------------------------------------------------------------------
// flat address of LDS offset 0 (the "shared aperture"); the host adds offsets to it to
// build DynArr.data pointers that point into a workgroup's LDS
__global__ void k_lds_aperture(unsigned long long* out)
{
    extern __shared__ __attribute__((aligned(16))) double dyn0[];
    *out = (unsigned long long)(void*)dyn0;
}

__global__ __launch_bounds__(WAVE) void k_ilqr_update(DevState st)
{
    __shared__ Lds s;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const Params* const P_mem = &st.params[b];
    Params P_regs;
    load_params_uniform(P_regs, P_mem);               // read-only for the launch: registers, not memory
    const Params& P = P_regs;
    double* sc = st.scalars + (int64_t)b * TPL_S_COUNT;

    const int T = (int)sc[TPL_S_HORIZON];
    const int opt_start = (int)sc[TPL_S_OPT_START];
    const int max_iter = (int)sc[TPL_S_MAX_ITERATIONS];
    const int max_lg = (int)sc[TPL_S_MAX_LG_ITERATIONS];
    const int int_type = (int)sc[TPL_S_INTEGRATOR_TYPE];
    const bool second_order = sc[TPL_S_USE_QUADRATIC_TERMS] != 0.0;
    const double dt = sc[TPL_S_DT];
    const double min_rel = sc[TPL_S_MIN_REL_COST_CHANGE];
    double mu = sc[TPL_S_MU];
    int mu_step = (int)sc[TPL_S_MU_STEP];
    double alpha = sc[TPL_S_ALPHA];
    int term = (int)sc[TPL_S_TERMINATION_CONDITION];
    bool improved = sc[TPL_S_IMPROVED] != 0.0;
    bool traj_changed = sc[TPL_S_TRAJECTORY_CHANGED] != 0.0;
    int iterations = (int)sc[TPL_S_ITERATIONS];
    int lg_it = 0;

    double* X_ = fld(st, TPL_F_X, b);
    double* U_ = fld(st, TPL_F_U, b);
    double* IS = fld(st, TPL_F_INT_STEP, b);
    double* NIS = fld(st, TPL_F_NEXT_INT_STEP, b);
    double* NXT_X = fld(st, TPL_F_NEXT_X, b);
    double* NXT_U = fld(st, TPL_F_NEXT_U, b);
    double* PRV_X = fld(st, TPL_F_PREV_X, b);
    double* PRV_K = fld(st, TPL_F_PREV_K, b);
    double* FX = fld(st, TPL_F_FX, b);
    double* FU = fld(st, TPL_F_FU, b);
    double* LX = fld(st, TPL_F_LX, b);
    double* LU = fld(st, TPL_F_LU, b);
    double* LXX = fld(st, TPL_F_LXX, b);
    double* LUU = fld(st, TPL_F_LUU, b);
    double* LUX = fld(st, TPL_F_LUX, b);
    double* G_ = fld(st, TPL_F_G, b);
    double* KFF = fld(st, TPL_F_K_FF, b);
    double* KFB = fld(st, TPL_F_K_FB, b);
    double* LAM = fld(st, TPL_F_LAGRANGE_MULTIPLIER, b);
    const double* BW = fld(st, TPL_F_BARRIER_WEIGHT, b);
    const double* LGL = fld(st, TPL_F_LG_MULT_LIMIT, b);
    const double* UMAX = fld(st, TPL_F_U_MAX, b);
    const double* UMIN = fld(st, TPL_F_U_MIN, b);
    double* CX = st.cand_x + (int64_t)b * N_ALPHA * st.rows * NX;
    double* CU = st.cand_u + (int64_t)b * N_ALPHA * st.rows * NU;

    // ---- LDS residency.  Array parameters: the host already pointed Params::<array>.data at
    // (aperture + offset); copy the data there.  The sanity check keeps a wrong aperture guess
    // from silently reading garbage: the problem is flagged and nothing is solved.
    if (st.n_lds_arrays > 0 || st.lds_traj) {
        if ((unsigned long long)(void*)dyn != st.lds_dyn_flat) {
            if (lane == 0) sc[TPL_S_TERMINATION_CONDITION] = -99.0;
            return;
        }
        for (int a = 0; a < st.n_lds_arrays; ++a) {
            const double* src = st.lds_src[a] + (int64_t)b * st.lds_src_stride[a];
            double* dst = dyn + st.lds_off[a];
            for (int e = lane; e < st.lds_len[a]; e += WAVE) dst[e] = src[e];
        }
    }
    // trajectory mirrors (generic pointers: LDS copy if it fits, else the global arrays)
    const int R = st.rows;
    double* XL = dyn;
    double* UL = XL + (int64_t)R * NX;
    double* KL = UL + (int64_t)R * NU;
    double* KBL = KL + (int64_t)R * NU;
    double* UMXL = KBL + (int64_t)R * NU * NX;
    double* UMNL = UMXL + (int64_t)R * NU;
    const bool lt = st.lds_traj != 0;
    const double* Xr = lt ? XL : X_;
    const double* Ur = lt ? UL : U_;
    const double* KFFr = lt ? KL : KFF;
    const double* KFBr = lt ? KBL : KFB;
    const double* UMAXr = lt ? UMXL : UMAX;
    const double* UMINr = lt ? UMNL : UMIN;
    if (lt) {
        for (int e = lane; e < T * NU; e += WAVE) { UMXL[e] = UMAX[e]; UMNL[e] = UMIN[e]; }
    }
    __syncthreads();

#ifdef TPL_ILQR_STAMP
    long long stamps[16] = {0};
    if (lane == 0 && st.stamps) for (int i = 4; i < 16; ++i) st.stamps[(int64_t)b * 16 + i] = 0;
    long long t_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    // ---- initial trajectory rollout and its cost, optim.c:1101-1111
    double* CC = st.cand_c + (int64_t)b * N_ALPHA * st.rows;
    double J = initial_rollout_cost(P, X_, U_, IS, LAM, BW, CC, T, opt_start, dt, int_type);
    J = __shfl(J, 0, WAVE);
    __syncthreads();
    if (lt) {
        for (int e = lane; e < (T + 1) * NX; e += WAVE) XL[e] = X_[e];
        for (int e = lane; e < T * NU; e += WAVE) UL[e] = U_[e];
    }
    __syncthreads();
    STAMP(0);

    for (lg_it = 0; lg_it < max_lg; ++lg_it) {
        multiplier_update(P, X_, U_, IS, LAM, BW, LGL, T, opt_start);      // optim.c:1119-1132
        __syncthreads();
        STAMP(1);
        traj_changed = true;
        improved = false;
        iterations = 0;

        // ---- ilqr() optim.c:875-1008 / ilr() :1010-1089
        for (int it = iterations; it < max_iter; ++it) {
            iterations = it + 1;

            if (traj_changed) {
                // derivative refresh, optim.c:896-911 (lane = stage)
                for (int t = lane; t < T; t += WAVE) {
                    double x[NX], u[NU], lam[NCM];
#pragma unroll
                    for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)t * NX + i];
#pragma unroll
                    for (int d = 0; d < NU; ++d) u[d] = U_[(int64_t)t * NU + d];
#pragma unroll
                    for (int c = 0; c < NCM; ++c) lam[c] = LAM[(int64_t)t * NCM + c];
                    const double ist = IS[t];
                    stateJacobian(P, x, u, (double)t, ist, FX + (int64_t)t * NX * NX, 1);
                    actionJacobian(P, x, u, (double)t, ist, FU + (int64_t)t * NX * NU, 1);
                    stateGradient(P, x, u, lam, BW, (double)t, ist, LX + (int64_t)t * NX, 1);
                    actionGradient(P, x, u, lam, BW, (double)t, ist, LU + (int64_t)t * NU, 1);
                    if (second_order) {
                        stateStateHessian(P, x, u, lam, BW, (double)t, ist, LXX + (int64_t)t * NX * NX, 1);
                        actionActionHessian(P, x, u, lam, BW, (double)t, ist, LUU + (int64_t)t * NU * NU, 1);
                        actionStateHessian(P, x, u, lam, BW, (double)t, ist, LUX + (int64_t)t * NU * NX, 1);
                    }
                }
                traj_changed = false;
                __syncthreads();
                STAMP(2);
            }

            // value function at the end of the horizon, optim.c:914-916
            if (lane == 0) {
                double x[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)T * NX + i];
                endGradient(P, x, (double)T, IS[T], s.Vx, 1);
                if (second_order) endHessian(P, x, (double)T, IS[T], s.Vxx, 1);
            }
            __syncthreads();

            if (second_order) {
                RiccatiArgs ra;
                ra.FX = FX; ra.LXX = LXX; ra.FU = FU; ra.LUX = LUX; ra.LX = LX; ra.LU = LU; ra.LUU = LUU;
                ra.U_ = U_; ra.UMAX = UMAX; ra.UMIN = UMIN; ra.KFF = KFF; ra.KFB = KFB; ra.KL = KL; ra.KBL = KBL;
                ra.mu = mu; ra.T = T; ra.lt = lt ? 1 : 0;
#ifdef TPL_ILQR_STAMP
                ra.stamp_acc = st.stamps ? st.stamps + (int64_t)b * 16 : nullptr;
#else
                ra.stamp_acc = nullptr;
#endif
#if TPL_ILQR_ABLATE != 1                    /* diagnostic builds: 1 = no backward pass, 2 = no rollouts */
                if constexpr (SMALL_BACKWARD) riccati_backward_small(ra, (LdsAS*)(unsigned long long)(unsigned)(unsigned long long)&s);
                else riccati_backward(ra, (LdsAS*)(unsigned long long)(unsigned)(unsigned long long)&s);
#endif
            } else {
                first_order_backward(s, FX, FU, LX, LU, U_, UMAX, UMIN, G_, KFF, KL, lt, T);   // optim.c:1044-1073
            }

            STAMP(3);
            // ---- line search, optim.c:859-873: all eight alphas at once, accept the first
            const double J_prev = J;
            // pass A: the eight dynamics chains (lqrForwardPass :745-775 / lrForwardPass :803-818)
            {
                RolloutArgs ra;
                ra.CX = CX; ra.CU = CU; ra.X_ = X_; ra.KFFr = KFFr; ra.Ur = Ur; ra.UMAXr = UMAXr; ra.UMINr = UMINr;
                ra.KFBr = KFBr; ra.Xr = Xr; ra.P = P_mem; ra.dt = dt; ra.rows = st.rows; ra.opt_start = opt_start;
                ra.T = T; ra.int_type = int_type; ra.second_order = second_order; ra.direct = st.roll_direct;
#ifdef TPL_ILQR_STAMP
                ra.stamp_acc = st.stamps ? st.stamps + (int64_t)b * 16 : nullptr;
#else
                ra.stamp_acc = nullptr;
#endif
                LdsAS* const sl = (LdsAS*)(unsigned long long)(unsigned)(unsigned long long)&s;
                auto roll = [&](auto lt_c, auto so_c) {
                    constexpr bool L = decltype(lt_c)::value, S2 = decltype(so_c)::value;
                    if (int_type == EULER) line_search_rollouts<L, S2, EULER>(ra, sl);
                    else if (int_type == HEUN) line_search_rollouts<L, S2, HEUN>(ra, sl);
                    else line_search_rollouts<L, S2, RK4>(ra, sl);
                };
#if TPL_ILQR_ABLATE != 2
                if (lt) { if (second_order) roll(std::true_type{}, std::true_type{}); else roll(std::true_type{}, std::false_type{}); }
                else { if (second_order) roll(std::false_type{}, std::true_type{}); else roll(std::false_type{}, std::false_type{}); }
#endif
            }
            __syncthreads();
            STAMP(11);
            IterTail tail;
            tail.X_ = X_; tail.U_ = U_; tail.IS = IS; tail.NIS = NIS; tail.NXT_X = NXT_X; tail.NXT_U = NXT_U;
            tail.PRV_X = PRV_X; tail.PRV_K = PRV_K; tail.KFF = KFF; tail.LAM = LAM; tail.BW = BW;
            tail.CX = CX; tail.CU = CU; tail.CC = CC; tail.XL = XL; tail.UL = UL;
            tail.rows = st.rows; tail.T = T; tail.opt_start = opt_start; tail.dt = dt; tail.lt = lt;
            const bool accepted = iteration_tail<true, false>(tail, P, s, J, alpha, traj_changed, improved, [&](int i) { (void)i; STAMP(i); });
            __syncthreads();

            STAMP(5);
            if (second_order) {                                // mu schedule, optim.c:989-999
                mu_step = accepted ? (mu_step - 1 > 0 ? mu_step - 1 : 0) : (mu_step + 1 < 7 ? mu_step + 1 : 7);
                if (mu_step == 0) {
                    mu = 0.0;
                } else {
                    double m = 1.0;
                    for (int i = 0; i < mu_step - 1; ++i) m *= 10.0;
                    mu = m;                                    // pow(10.0, muStep - 1)
                }
            }
            const double rel = fabs(J - J_prev) / J;
            if (rel < min_rel) {                               // optim.c:1001-1006
                term = 2;
                break;
            }
        }
    }
    if (iterations == max_iter) term = 1;                      // optim.c:1147-1149

#ifdef TPL_ILQR_STAMP
    if (lane == 0) for (int i = 0; i < 16; ++i) if (i < 4 || i == 5 || (i > 10 && i < 14)) st.stamps[(int64_t)b * 16 + i] = stamps[i];
#endif
    if (lane == 0) {
        sc[TPL_S_TRAJ_COSTS] = J;
        sc[TPL_S_ITERATIONS] = (double)iterations;
        sc[TPL_S_LG_ITERATIONS] = (double)lg_it;
        sc[TPL_S_ALPHA] = alpha;
        sc[TPL_S_MU] = mu;
        sc[TPL_S_MU_STEP] = (double)mu_step;
        sc[TPL_S_TRAJECTORY_CHANGED] = traj_changed ? 1.0 : 0.0;
        sc[TPL_S_IMPROVED] = improved ? 1.0 : 0.0;
        sc[TPL_S_TERMINATION_CONDITION] = (double)term;
    }
}

// ---------------------------------------------------------------------------
// The same update() as a SEQUENCE OF LAUNCHES, for batches that fill the chip more than once.
//
// k_ilqr_update keeps the whole solve in one launch, which is what a latency caller wants -- but a
// wavefront's register allocation is fixed at launch to the maximum over the kernel's call graph, and
// the derivative refresh (hundreds of live temporaries) drives that to 484 registers + scratch: ONE
// wavefront per SIMD, every dependent fp64 chain of the Riccati pass and of the rollouts exposed in
// full.  Splitting the iteration by register class gives the chains a kernel of their own:
//
//   k_split_begin    initial rollout and cost (first round only) + multiplier update   optim.c:1101-1132
//   k_split_refresh  derivative refresh, lane = stage, any register count            optim.c:896-911
//   k_split_chain    terminal value function, Riccati pass, the eight rollouts, their stage costs,
//                    accept-first, mu schedule, termination test                     optim.c:914-1006
//                    -- no derivative routine in its call graph: it is built for >= 2 resident
//                    wavefronts per SIMD (TPL_ILQR_CHAIN_WAVES), which hide each other's chains
//   k_split_end      terminationCondition = 1 when the iteration budget was used up  optim.c:1147-1149
//
// The state that the single launch carries in registers between iterations (J, mu, muStep, alpha, the
// flags, the iteration counter) lives in the problem's scalar record between launches; a problem that
// left its iteration loop (min_rel_cost_change) is marked in `run` and its later launches return at
// once.  Same routines, same operation order: both forms produce the same bits
// (tests/test_ilqr_gpu.py::test_split_launch_matches_single_launch).
// ---------------------------------------------------------------------------
#ifndef TPL_ILQR_CHAIN_WAVES
#define TPL_ILQR_CHAIN_WAVES 2
#endif
#ifndef TPL_ILQR_CHAIN_WAVES_MAX
#define TPL_ILQR_CHAIN_WAVES_MAX TPL_ILQR_CHAIN_WAVES
#endif
#ifndef TPL_ILQR_CHAIN_INLINE    /* 1: the Riccati pass and the rollouts inlined into k_split_chain (its register budget then binds them) */
#define TPL_ILQR_CHAIN_INLINE 1
#endif
constexpr bool CHAIN_INLINE = TPL_ILQR_CHAIN_INLINE != 0;
constexpr int LDS_HEAD = (int)((sizeof(Lds) + 15) / 16 * 2);      // doubles the Lds block takes at the head of the dynamic LDS

struct SplitProb {
    double* sc;
    int T, opt_start, max_iter, max_lg, int_type;
    bool second_order;
    double dt;
};

// The split kernels and the fused kernel below read the device state through a pointer (scalar loads from the constant
// address space, field by field on demand): their phases are real functions, and a 600-byte struct passed by value would
// travel through every lane's stack at each call.
typedef const __attribute__((address_space(4))) DevState* CstDev;
__device__ __forceinline__ CstDev uni_dev(const DevState* p) { return (CstDev)uni_u64((unsigned long long)p); }

__device__ __forceinline__ double* fld(CstDev st, int f, int b)
{
    const int64_t rows = field_is_per_stage(f) ? st->rows : 1;
    return (double*)st->f[f] + (int64_t)b * rows * field_width(f);
}

__device__ __forceinline__ SplitProb split_prob(CstDev st, int b)
{
    SplitProb p;
    p.sc = (double*)st->scalars + (int64_t)b * TPL_S_COUNT;
    p.T = (int)p.sc[TPL_S_HORIZON];
    p.opt_start = (int)p.sc[TPL_S_OPT_START];
    p.max_iter = (int)p.sc[TPL_S_MAX_ITERATIONS];
    p.max_lg = (int)p.sc[TPL_S_MAX_LG_ITERATIONS];
    p.int_type = (int)p.sc[TPL_S_INTEGRATOR_TYPE];
    p.second_order = p.sc[TPL_S_USE_QUADRATIC_TERMS] != 0.0;
    p.dt = p.sc[TPL_S_DT];
    return p;
}

// array parameters -> LDS (the host pointed Params::<array>.data there); false: wrong aperture guess
__device__ __forceinline__ bool split_lds_arrays(CstDev st, int b, double* dyn, double* sc)
{
    const int lane = threadIdx.x;
    const int n_arrays = st->n_lds_arrays;
    if (n_arrays > 0 || st->lds_traj) {
        if ((unsigned long long)(void*)dyn != st->lds_dyn_flat) {
            if (lane == 0) sc[TPL_S_TERMINATION_CONDITION] = -99.0;
            return false;
        }
        for (int a = 0; a < n_arrays; ++a) {
            const double* src = (const double*)st->lds_src[a] + (int64_t)b * st->lds_src_stride[a];
            double* dst = dyn + st->lds_off[a];
            const int len = st->lds_len[a];
            for (int e = lane; e < len; e += WAVE) dst[e] = src[e];
        }
    }
    return true;
}

// FUSED (all three phases): the caller is k_ilqr_fused, whose LDS lives for the whole update -- the array parameters and
// the mirrors of the control bounds were brought in once by fused_setup() and are not fetched again by every phase.
template <bool FUSED>
__device__ __forceinline__ void ph_begin(const DevState* stp, int lg_in, int b_in)
{
    const CstDev st = uni_dev(stp);
    double* const dynall = (double*)st->lds_base_flat;         // (a function cannot name the kernel's dynamic LDS block: see k_split_begin)
    double* const dyn = dynall + LDS_HEAD;
    const int lg = uni_i(lg_in), b = uni_i(b_in);
    const int lane = threadIdx.x;
    const SplitProb pr = split_prob(st, b);
    double* const sc = pr.sc;
    if (lg > 0 && lg >= pr.max_lg) return;
    Params P_regs;
    load_params_uniform(P_regs, (const Params*)st->params + b);
    const Params& P = P_regs;
    const int T = pr.T, opt_start = pr.opt_start, int_type = pr.int_type;
    const double dt = pr.dt;
    double* X_ = fld(st, TPL_F_X, b);
    double* U_ = fld(st, TPL_F_U, b);
    double* IS = fld(st, TPL_F_INT_STEP, b);
    double* LAM = fld(st, TPL_F_LAGRANGE_MULTIPLIER, b);
    const double* BW = fld(st, TPL_F_BARRIER_WEIGHT, b);
    const double* LGL = fld(st, TPL_F_LG_MULT_LIMIT, b);
    if constexpr (!FUSED) {
        if (!split_lds_arrays(st, b, dyn, sc)) return;
        __syncthreads();
    }

    if (lg == 0) {
        // ---- initial trajectory rollout and its cost, optim.c:1101-1111
        double* CC = (double*)st->cand_c + (int64_t)b * N_ALPHA * st->rows;
        const double J = initial_rollout_cost(P, X_, U_, IS, LAM, BW, CC, T, opt_start, dt, int_type);
        if (lane == 0) {
            sc[TPL_S_TRAJ_COSTS] = J;
            sc[TPL_S_LG_ITERATIONS] = (double)(pr.max_lg > 0 ? pr.max_lg : 0);
        }
        __syncthreads();
        if (pr.max_lg <= 0) return;
    }
    // ---- multiplier update, optim.c:1119-1132, and the loop state of this round
    multiplier_update(P, X_, U_, IS, LAM, BW, LGL, T, opt_start);
    if (lane == 0) {
        sc[TPL_S_TRAJECTORY_CHANGED] = 1.0;
        sc[TPL_S_IMPROVED] = 0.0;
        sc[TPL_S_ITERATIONS] = 0.0;
        ((int*)st->run)[b] = 0;
    }
    if constexpr (FUSED) {
        // the trajectory's LDS mirrors: filled here (after the initial rollout), kept current by the chain phase when it
        // accepts a step, read by the refresh and the chain -- the split sequence refills them in every chain launch
        if (st->lds_traj) {
            __syncthreads();
            double* XL = dyn;
            double* UL = XL + (int64_t)st->rows * NX;
            for (int e = lane; e < (T + 1) * NX; e += WAVE) XL[e] = X_[e];
            for (int e = lane; e < T * NU; e += WAVE) UL[e] = U_[e];
        }
    }
}

// Derivative refresh: one wavefront per problem, lane = stage, as in k_ilqr_update.  (Measured and dropped,
// profiles/r04_ilqr_split_ab.txt: the seven routines spread over three wavefronts per problem -- Jacobians / gradients
// and small Hessians / state Hessian -- 3.8 -> 4.8 ms per update of 1 024 problems: a wavefront's fixed start-up, the
// parameter record, the arrays' copy to LDS, the state loads, costs more than the routines it no longer runs; and the
// kernel held to 256 registers for two wavefronts per SIMD: 218 registers in scratch, 10.9 -> 12.1 ms at 4 096.)
template <bool FUSED>
__device__ __forceinline__ void ph_refresh_block(const DevState* stp, int lg_in, int it_in, int b_in, int block_in)
{
    const CstDev st = uni_dev(stp);
    double* const dynall = (double*)st->lds_base_flat;
    double* const dyn = dynall + LDS_HEAD;
    const int lg = uni_i(lg_in), it = uni_i(it_in), b = uni_i(b_in), block = uni_i(block_in);
    const int lane = threadIdx.x;
    const SplitProb pr = split_prob(st, b);
    if (lg >= pr.max_lg || it >= pr.max_iter || ((int*)st->run)[b] != 0) return;
    if (pr.sc[TPL_S_TRAJECTORY_CHANGED] == 0.0) return;
    const bool second_order = pr.second_order;
    Params P_regs;
    load_params_uniform(P_regs, (const Params*)st->params + b);
    const Params& P = P_regs;
    if constexpr (!FUSED) {
        if (!split_lds_arrays(st, b, dyn, pr.sc)) return;
        __syncthreads();
    }
    const int T = pr.T;
    const double* X_ = fld(st, TPL_F_X, b);
    const double* U_ = fld(st, TPL_F_U, b);
    if constexpr (FUSED) {
        if (st->lds_traj) { X_ = dyn; U_ = dyn + (int64_t)st->rows * NX; }       // the mirrors (ph_begin, ph_chain keep them current)
    }
    const double* IS = fld(st, TPL_F_INT_STEP, b);
    const double* LAM = fld(st, TPL_F_LAGRANGE_MULTIPLIER, b);
    const double* BW = fld(st, TPL_F_BARRIER_WEIGHT, b);
    double* FX = fld(st, TPL_F_FX, b);
    double* FU = fld(st, TPL_F_FU, b);
    double* LX = fld(st, TPL_F_LX, b);
    double* LU = fld(st, TPL_F_LU, b);
    double* LXX = fld(st, TPL_F_LXX, b);
    double* LUU = fld(st, TPL_F_LUU, b);
    double* LUX = fld(st, TPL_F_LUX, b);
    // blockIdx.y = block of 64 stages: NO loop over the stages in here, and one routine at a time (the largest first,
    // nothing scheduled across the boundaries).  Around a stage loop the compiler hoists every parameter-only
    // sub-expression of the seven routines and keeps them alive across all of them: 385 registers + 112 parked in AGPRs;
    // this form holds 276, nothing parked.  (Each routine alone fits 98 ... 252 registers on the MPC model.  Measured with
    // it, profiles/r04_ilqr_split_ab.txt: a second resident refresh wavefront -- 256 registers, 24 spilled -- does not
    // help, nor do rows staged through LDS for contiguous stores: neither occupancy nor the strided stores bind it.)
    {
        const int t = block * WAVE + lane;
        if (t >= T) return;
        double x[NX], u[NU], lam[NCM];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = X_[(int64_t)t * NX + i];
#pragma unroll
        for (int d = 0; d < NU; ++d) u[d] = U_[(int64_t)t * NU + d];
#pragma unroll
        for (int c = 0; c < NCM; ++c) lam[c] = LAM[(int64_t)t * NCM + c];
        const double ist = IS[t];
        if (second_order) stateStateHessian(P, x, u, lam, BW, (double)t, ist, LXX + (int64_t)t * NX * NX, 1);
        __builtin_amdgcn_sched_barrier(0);
        stateGradient(P, x, u, lam, BW, (double)t, ist, LX + (int64_t)t * NX, 1);
        __builtin_amdgcn_sched_barrier(0);
        stateJacobian(P, x, u, (double)t, ist, FX + (int64_t)t * NX * NX, 1);
        __builtin_amdgcn_sched_barrier(0);
        actionJacobian(P, x, u, (double)t, ist, FU + (int64_t)t * NX * NU, 1);
        actionGradient(P, x, u, lam, BW, (double)t, ist, LU + (int64_t)t * NU, 1);
        if (second_order) {
            actionActionHessian(P, x, u, lam, BW, (double)t, ist, LUU + (int64_t)t * NU * NU, 1);
            actionStateHessian(P, x, u, lam, BW, (double)t, ist, LUX + (int64_t)t * NU * NX, 1);
        }
    }
}

template <bool SECOND_ORDER, int INT_TYPE, bool FUSED>
__device__ __forceinline__ void ph_chain(const DevState* stp, int lg_in, int it_in, int b_in)
{
    const CstDev st = uni_dev(stp);
    double* const dynall = (double*)st->lds_base_flat;
    Lds& s = *reinterpret_cast<Lds*>(dynall);
    double* const dyn = dynall + LDS_HEAD;
    const int lg = uni_i(lg_in), it = uni_i(it_in), b = uni_i(b_in);
    const int lane = threadIdx.x;
    const SplitProb pr = split_prob(st, b);
    double* const sc = pr.sc;
    if (lg >= pr.max_lg || it >= pr.max_iter || ((int*)st->run)[b] != 0) return;
    const Params* const P_mem = (const Params*)st->params + b;
    Params P_regs;
    load_params_uniform(P_regs, P_mem);
    const Params& P = P_regs;
    const int T = pr.T, opt_start = pr.opt_start;
    const double dt = pr.dt;
    const double min_rel = sc[TPL_S_MIN_REL_COST_CHANGE];
    double mu = sc[TPL_S_MU];
    int mu_step = (int)sc[TPL_S_MU_STEP];
    double alpha;
    double J = sc[TPL_S_TRAJ_COSTS];
    bool improved = sc[TPL_S_IMPROVED] != 0.0;
    bool traj_changed = false;                         // k_split_refresh has just brought the derivatives up to date

    double* X_ = fld(st, TPL_F_X, b);
    double* U_ = fld(st, TPL_F_U, b);
    double* IS = fld(st, TPL_F_INT_STEP, b);
    double* NIS = fld(st, TPL_F_NEXT_INT_STEP, b);
    double* NXT_X = fld(st, TPL_F_NEXT_X, b);
    double* NXT_U = fld(st, TPL_F_NEXT_U, b);
    double* PRV_X = fld(st, TPL_F_PREV_X, b);
    double* PRV_K = fld(st, TPL_F_PREV_K, b);
    double* FX = fld(st, TPL_F_FX, b);
    double* FU = fld(st, TPL_F_FU, b);
    double* LX = fld(st, TPL_F_LX, b);
    double* LU = fld(st, TPL_F_LU, b);
    double* LXX = fld(st, TPL_F_LXX, b);
    double* LUU = fld(st, TPL_F_LUU, b);
    double* LUX = fld(st, TPL_F_LUX, b);
    double* G_ = fld(st, TPL_F_G, b);
    double* KFF = fld(st, TPL_F_K_FF, b);
    double* KFB = fld(st, TPL_F_K_FB, b);
    double* LAM = fld(st, TPL_F_LAGRANGE_MULTIPLIER, b);
    const double* BW = fld(st, TPL_F_BARRIER_WEIGHT, b);
    const double* UMAX = fld(st, TPL_F_U_MAX, b);
    const double* UMIN = fld(st, TPL_F_U_MIN, b);
    double* CX = (double*)st->cand_x + (int64_t)b * N_ALPHA * st->rows * NX;
    double* CU = (double*)st->cand_u + (int64_t)b * N_ALPHA * st->rows * NU;
    double* CC = (double*)st->cand_c + (int64_t)b * N_ALPHA * st->rows;

    if constexpr (!FUSED) {
        if (!split_lds_arrays(st, b, dyn, sc)) return;
    }
    const int R = st->rows;
    double* XL = dyn;
    double* UL = XL + (int64_t)R * NX;
    double* KL = UL + (int64_t)R * NU;
    double* KBL = KL + (int64_t)R * NU;
    double* UMXL = KBL + (int64_t)R * NU * NX;
    double* UMNL = UMXL + (int64_t)R * NU;
    const bool lt = st->lds_traj != 0;
    const double* Xr = lt ? XL : X_;
    const double* Ur = lt ? UL : U_;
    const double* KFFr = lt ? KL : KFF;
    const double* KFBr = lt ? KBL : KFB;
    const double* UMAXr = lt ? UMXL : UMAX;
    const double* UMINr = lt ? UMNL : UMIN;
    if constexpr (!FUSED) {
        if (lt) {                                      // the mirrors live for one launch: bring them in
            for (int e = lane; e < T * NU; e += WAVE) { UMXL[e] = UMAX[e]; UMNL[e] = UMIN[e]; UL[e] = U_[e]; }
            for (int e = lane; e < (T + 1) * NX; e += WAVE) XL[e] = X_[e];
        }
    }                                                  // (fused: bounds from fused_setup, x and u from ph_begin / the last accepted step)
    // value function at the end of the horizon, optim.c:914-916
    if (lane == 0) {
        double x[NX];
        const double* xs = (FUSED && lt) ? XL : X_;           // (fused: the mirror is current, and nearer)
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xs[(int64_t)T * NX + i];
        endGradient(P, x, (double)T, IS[T], s.Vx, 1);
        if (SECOND_ORDER) endHessian(P, x, (double)T, IS[T], s.Vxx, 1);
    }
    __syncthreads();

    // The Lds block sits at the head of the dynamic LDS: LDS offset = the low half of its flat address.
    const unsigned sl_off = (unsigned)uni_i((int)(unsigned)st->lds_base_flat);
    LdsAS* const sl = (LdsAS*)(unsigned long long)sl_off;
    if constexpr (SECOND_ORDER) {
        RiccatiArgs ra;
        ra.FX = FX; ra.LXX = LXX; ra.FU = FU; ra.LUX = LUX; ra.LX = LX; ra.LU = LU; ra.LUU = LUU;
        ra.U_ = U_; ra.UMAX = UMAX; ra.UMIN = UMIN; ra.KFF = KFF; ra.KFB = KFB; ra.KL = KL; ra.KBL = KBL;
        ra.mu = mu; ra.T = T; ra.lt = lt ? 1 : 0;
        ra.stamp_acc = nullptr;
        if constexpr (SMALL_BACKWARD) riccati_backward_small(ra, sl);
        else if constexpr (CHAIN_INLINE) riccati_backward_impl(ra, sl);
        else riccati_backward(ra, sl);
    } else {
        first_order_backward(s, FX, FU, LX, LU, U_, UMAX, UMIN, G_, KFF, KL, lt, T);   // optim.c:1044-1073
    }

    // ---- line search, optim.c:859-873: all eight alphas at once, accept the first
    const double J_prev = J;
    {
        RolloutArgs ra;
        ra.CX = CX; ra.CU = CU; ra.X_ = X_; ra.KFFr = KFFr; ra.Ur = Ur; ra.UMAXr = UMAXr; ra.UMINr = UMINr;
        ra.KFBr = KFBr; ra.Xr = Xr; ra.P = P_mem; ra.dt = dt; ra.rows = st->rows; ra.opt_start = opt_start;
        ra.T = T; ra.int_type = INT_TYPE; ra.second_order = SECOND_ORDER; ra.direct = st->roll_direct;
        ra.stamp_acc = nullptr;
        if constexpr (CHAIN_INLINE) {
            if (lt) line_search_rollouts_impl<true, SECOND_ORDER, INT_TYPE>(ra, sl);
            else line_search_rollouts_impl<false, SECOND_ORDER, INT_TYPE>(ra, sl);
        } else {
            if (lt) line_search_rollouts<true, SECOND_ORDER, INT_TYPE>(ra, sl);
            else line_search_rollouts<false, SECOND_ORDER, INT_TYPE>(ra, sl);
        }
    }
    __syncthreads();
    IterTail tail;
    tail.X_ = X_; tail.U_ = U_; tail.IS = IS; tail.NIS = NIS; tail.NXT_X = NXT_X; tail.NXT_U = NXT_U;
    tail.PRV_X = PRV_X; tail.PRV_K = PRV_K; tail.KFF = KFF; tail.LAM = LAM; tail.BW = BW;
    tail.CX = CX; tail.CU = CU; tail.CC = CC; tail.XL = XL; tail.UL = UL;
    tail.rows = st->rows; tail.T = T; tail.opt_start = opt_start; tail.dt = dt; tail.lt = lt;
    const bool accepted = iteration_tail<FUSED, FUSED>(tail, P, s, J, alpha, traj_changed, improved, [](int) {});
    if (SECOND_ORDER) {                                // mu schedule, optim.c:989-999
        mu_step = accepted ? (mu_step - 1 > 0 ? mu_step - 1 : 0) : (mu_step + 1 < 7 ? mu_step + 1 : 7);
        if (mu_step == 0) {
            mu = 0.0;
        } else {
            double m = 1.0;
            for (int i = 0; i < mu_step - 1; ++i) m *= 10.0;
            mu = m;
        }
    }
    const double rel = fabs(J - J_prev) / J;
    if (lane == 0) {
        sc[TPL_S_TRAJ_COSTS] = J;
        sc[TPL_S_ITERATIONS] = (double)(it + 1);
        sc[TPL_S_ALPHA] = alpha;
        sc[TPL_S_MU] = mu;
        sc[TPL_S_MU_STEP] = (double)mu_step;
        sc[TPL_S_TRAJECTORY_CHANGED] = traj_changed ? 1.0 : 0.0;
        sc[TPL_S_IMPROVED] = improved ? 1.0 : 0.0;
        if (rel < min_rel) {                           // optim.c:1001-1006: leaves the iteration loop of this round
            sc[TPL_S_TERMINATION_CONDITION] = 2.0;
            ((int*)st->run)[b] = 1;
        }
    }
}

// ---- the launches of the split sequence: one phase each
// (every kernel first checks that its dynamic LDS block is where the host assumed -- LDS offset 0 behind the shared
// aperture -- and flags the problem instead of computing on garbage if not)
__device__ __forceinline__ bool lds_where_expected(const DevState* stp)
{
    extern __shared__ __attribute__((aligned(16))) double dynall[];
    if ((unsigned long long)(void*)dynall == stp->lds_base_flat) return true;
    if (threadIdx.x == 0) stp->scalars[(int64_t)blockIdx.x * TPL_S_COUNT + TPL_S_TERMINATION_CONDITION] = -99.0;
    return false;
}

__global__ __launch_bounds__(WAVE) void k_split_begin(const DevState* stp, int lg) { if (lds_where_expected(stp)) ph_begin<false>(stp, lg, (int)blockIdx.x); }

__global__ __launch_bounds__(WAVE) void k_split_refresh(const DevState* stp, int lg, int it)
{
    if (lds_where_expected(stp)) ph_refresh_block<false>(stp, lg, it, (int)blockIdx.x, (int)blockIdx.y);
}

template <bool SECOND_ORDER, int INT_TYPE>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(TPL_ILQR_CHAIN_WAVES, TPL_ILQR_CHAIN_WAVES_MAX)))
void k_split_chain(const DevState* stp, int lg, int it)
{
    if (lds_where_expected(stp)) ph_chain<SECOND_ORDER, INT_TYPE, false>(stp, lg, it, (int)blockIdx.x);
}

// ---- ... and the SAME phases behind ONE launch: k_ilqr_fused.  What kept k_ilqr_update at one wavefront per SIMD was not
// the sum of its phases but the compiler's view of them as one body -- parameter-only sub-expressions hoisted out of the
// iteration and stage loops and kept alive across everything (484 registers + scratch).  As real calls every phase is
// compiled by itself, to the 256 registers of a two-wavefront budget (MPC model: begin 248, refresh block 256 with ~75
// spilled, chain 248; each saves the callee-saved registers it uses on entry), and the kernel holds two wavefronts per
// SIMD with no launch between the phases; the loop state travels through the problem's scalar record as in the split
// sequence (a few loads per phase, workgroup fences between phases).  Same routines, same order: the same bits as both
// other forms.  8.5-8.7 M problem-iterations/s at 4 096 problems against 6.7 (single) / 7.8 (split):
// profiles/r04_ilqr_split_ab.txt.
// (the calls are to these out-of-line copies, which only k_ilqr_fused uses: a callee is compiled to the register budget
// its callers share, so they get the two-wavefront budget while the split kernels above keep their own)
// (W = wavefronts per SIMD the calling kernel is compiled for: one set of out-of-line copies per instance of the kernel, so
// that each set is compiled to ITS caller's register budget -- 256 registers under W = 2, 512 under W = 1)
template <int W> __device__ __attribute__((noinline)) void fn_begin(const DevState* stp, int lg, int b) { ph_begin<true>(stp, lg, b); }
template <int W> __device__ __attribute__((noinline)) void fn_refresh_block(const DevState* stp, int lg, int it, int b, int blk) { ph_refresh_block<true>(stp, lg, it, b, blk); }
template <bool SECOND_ORDER, int INT_TYPE, int W>
__device__ __attribute__((noinline)) void fn_chain(const DevState* stp, int lg, int it, int b) { ph_chain<SECOND_ORDER, INT_TYPE, true>(stp, lg, it, b); }

// what every phase of a launch of the split sequence fetches for itself, once per update: the array parameters and the
// control bounds' mirrors into the LDS regions all phases agree on
template <int W>
__device__ __attribute__((noinline)) void fused_setup(const DevState* stp, int b_in)
{
    const CstDev st = uni_dev(stp);
    double* const dyn = (double*)st->lds_base_flat + LDS_HEAD;
    const int b = uni_i(b_in), lane = threadIdx.x;
    const SplitProb pr = split_prob(st, b);
    if (!split_lds_arrays(st, b, dyn, pr.sc)) return;
    if (st->lds_traj) {
        const int R = st->rows;
        double* UMXL = dyn + (int64_t)R * (NX + NU + NU + NU * NX);
        double* UMNL = UMXL + (int64_t)R * NU;
        const double* UMAX = fld(st, TPL_F_U_MAX, b);
        const double* UMIN = fld(st, TPL_F_U_MIN, b);
        for (int e = lane; e < pr.T * NU; e += WAVE) { UMXL[e] = UMAX[e]; UMNL[e] = UMIN[e]; }
    }
}

template <bool SECOND_ORDER, int INT_TYPE, int W>
__device__ __forceinline__ void fused_body(const DevState* stp, int rounds, int iterations, int blocks)
{
    const int b = (int)blockIdx.x;
    if (!lds_where_expected(stp)) return;
    fused_setup<W>(stp, b);
    __syncthreads();
    for (int lg = 0; lg < rounds; ++lg) {
        fn_begin<W>(stp, lg, b);
        __syncthreads();
        for (int it = 0; it < iterations; ++it) {
            for (int blk = 0; blk < blocks; ++blk) fn_refresh_block<W>(stp, lg, it, b, blk);
            __syncthreads();
            fn_chain<SECOND_ORDER, INT_TYPE, W>(stp, lg, it, b);
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        double* sc = stp->scalars + (int64_t)b * TPL_S_COUNT;
        if (sc[TPL_S_TERMINATION_CONDITION] != -99.0 && (int)sc[TPL_S_ITERATIONS] == (int)sc[TPL_S_MAX_ITERATIONS])
            sc[TPL_S_TERMINATION_CONDITION] = 1.0;                   // optim.c:1147-1149
    }
}

template <bool SECOND_ORDER, int INT_TYPE>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(TPL_ILQR_CHAIN_WAVES, TPL_ILQR_CHAIN_WAVES_MAX)))
void k_ilqr_fused(const DevState* stp, int rounds, int iterations, int blocks)
{
    fused_body<SECOND_ORDER, INT_TYPE, TPL_ILQR_CHAIN_WAVES>(stp, rounds, iterations, blocks);
}

// The same phases as calls at ONE wavefront per SIMD (round 5): what k_ilqr_update is for batches of up to 4 x CUs problems,
// without the one body -- every phase compiled by itself to the full register file (no AGPR parking, no scratch), nothing
// hoisted across a call.  TPLHIP_DEBUG=ilqr_split=3 selects it; whether update() takes it by itself: pick_form().
template <bool SECOND_ORDER, int INT_TYPE>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_ilqr_fused1(const DevState* stp, int rounds, int iterations, int blocks)
{
    fused_body<SECOND_ORDER, INT_TYPE, 1>(stp, rounds, iterations, blocks);
}

__global__ void k_split_end(const DevState* stp, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double* sc = stp->scalars + (int64_t)b * TPL_S_COUNT;
    if (sc[TPL_S_TERMINATION_CONDITION] == -99.0) return;
    if ((int)sc[TPL_S_ITERATIONS] == (int)sc[TPL_S_MAX_ITERATIONS]) sc[TPL_S_TERMINATION_CONDITION] = 1.0;   // optim.c:1147-1149
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

thread_local char g_error[1024] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr, context)                                                              \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) return fail(TPLHIP_EDEVICE, "%s: %s", hipGetErrorString(e__), context); \
    } while (0)

struct ParamRef { const char* name; int is_array; size_t offset; };
#define TPL_PARAM_S(n) {#n, 0, offsetof(Params, n)},
#define TPL_PARAM_A(n) {#n, 1, offsetof(Params, n)},
const ParamRef g_params[] = { TPL_PARAM_LIST {nullptr, 0, 0} };
#undef TPL_PARAM_S
#undef TPL_PARAM_A

struct ArrayParam {
    std::vector<double> host;        // [batch or 1] x elems
    int ndim = 0;
    int64_t dims[8] = {0};
    int64_t elems = 0;               // per problem
    int per_problem = 0;
    double* dev = nullptr;           // into the handle's device parameter arena (not owned)
    long long lds_off = -1;          // doubles from the dynamic LDS base, -1: stays in global memory
    long long lds_off_split = -1;    // the same for the split-launch kernels (their own LDS layout)
};

// fields the kernel reads from the host's hands, and the ones update() returns eagerly
const int kInputFields[] = {TPL_F_X, TPL_F_U, TPL_F_LAGRANGE_MULTIPLIER, TPL_F_BARRIER_WEIGHT,
                            TPL_F_LG_MULT_LIMIT, TPL_F_U_MAX, TPL_F_U_MIN};
const int kEagerFields[] = {TPL_F_INT_STEP, TPL_F_X, TPL_F_U, TPL_F_NEXT_INT_STEP, TPL_F_NEXT_X, TPL_F_NEXT_U,
                            TPL_F_PREV_X, TPL_F_PREV_K, TPL_F_K_FF, TPL_F_K_FB, TPL_F_LAGRANGE_MULTIPLIER, TPL_F_G};

// One arena on each side, same offsets: the host mirrors are page-locked and ordered so that
// everything update() hands to the kernel (input-only fields, the scalars, x / u / lambda) is ONE
// contiguous range and everything it returns eagerly (the scalars, x / u / lambda, the rest of
// kEagerFields) is another -- one copy in, one copy out, instead of one per field from pageable
// memory (20 copies of a few KB each were 40 % of a batch-1 MPC tick).  -1 marks the scalars.
const int kArenaOrder[] = {TPL_F_BARRIER_WEIGHT, TPL_F_LG_MULT_LIMIT, TPL_F_U_MAX, TPL_F_U_MIN,
                           -1, TPL_F_X, TPL_F_U, TPL_F_LAGRANGE_MULTIPLIER,
                           TPL_F_INT_STEP, TPL_F_NEXT_INT_STEP, TPL_F_NEXT_X, TPL_F_NEXT_U, TPL_F_PREV_X,
                           TPL_F_PREV_K, TPL_F_K_FF, TPL_F_K_FB, TPL_F_G};
const size_t kArenaAlign = 32;       // doubles: every segment starts on a 256-byte boundary

struct Span {                        // a segment of the host arena
    double* p = nullptr;
    size_t n = 0;
    double* data() const { return p; }
    size_t size() const { return n; }
    double& operator[](size_t i) const { return p[i]; }
};

} // namespace

struct tplhip_ilqr {
    int device = 0;
    int batch = 1;
    int t_cap = 0;
    int rows = 0;
    hipStream_t stream = nullptr;
    double* h_arena = nullptr;                   // page-locked host mirrors, layout kArenaOrder
    double* d_arena = nullptr;                   // the same layout on the device
    size_t arena_elems = 0;
    size_t in_lo = 0, in_hi = 0;                 // [lo, hi) doubles: what update() uploads
    size_t out_lo = 0, out_hi = 0;               // ... and what it brings back
    Span host[TPL_F_COUNT];
    bool device_newer[TPL_F_COUNT] = {false};
    Span scalars;                                // [batch][S_COUNT]
    std::vector<std::vector<double>> pscalar;    // [n_params][batch]
    std::vector<ArrayParam> parray;              // [n_params]
    std::vector<Params> hparams;                 // [batch], host-pointer version for dynamics()
    char* h_parena = nullptr;                    // page-locked staging: [Params x batch][array parameters ...]
    char* d_parena = nullptr;                    // its device twin; d_params = its head
    size_t parena_cap = 0;
    DevState dev{};
    Params* d_params = nullptr;
    double* d_snapshot[3] = {nullptr, nullptr, nullptr};   // x, u, lambda
    double* d_snapshot_scalars = nullptr;
    bool have_snapshot = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    bool timing = false;
    double kernel_ms = 0.0;
    int64_t launches = 0;
    int n_cu = 256;                              // compute units of the device
    unsigned long long lds_aperture = 0;         // flat address of LDS offset 0 (0: unknown, no LDS residency)
    size_t lds_static = 0;                       // static LDS of k_ilqr_update
    size_t lds_dyn_bytes = 0;                    // dynamic LDS of the next launch
    // split-launch form (k_split_*): its own view of the device state (parameter records that point into ITS LDS layout)
    DevState dev_split{};
    DevState* h_dev_split = nullptr;             // page-locked staging of dev_split ...
    DevState* d_dev_split = nullptr;             // ... and its copy in device memory: what the split / fused kernels read
    DevState dev_split_uploaded{};               // last content sent (re-sent only when it changes)
    bool dev_split_valid = false;
    Params* d_params_split = nullptr;
    size_t lds_dyn_bytes_split = 0;              // dynamic LDS of the split kernels (Lds block + arrays + mirrors)
    int* d_run = nullptr;
    int split_mode = -1;                         // TPLHIP_DEBUG ilqr_split=0|1|2; -1: by batch size and model
    bool split_uniform = false;                  // integrator and order are the same for the whole batch
    int split_int_type = 0;
    bool split_second_order = true;
    int max_iter_max = 0, max_lg_max = 0;        // over the batch: how many launches a round / an update needs
    int horizon_max = 1;                         // ... and how many blocks of 64 stages the refresh covers
    int last_form = 0;                           // 0 single launch, 1 split (info)
};

namespace {

size_t field_elems(const tplhip_ilqr* h, int f)
{
    const size_t rows = field_is_per_stage(f) ? (size_t)h->rows : 1;
    return (size_t)h->batch * rows * field_width(f);
}

// Decide what lives in LDS for the next launch.  Batches that fill the chip keep one workgroup per SIMD resident
// (4 per CU, 160 KB of LDS per CU), so a workgroup may use ~36 KB on top of its static block; a SMALL batch -- the
// one problem of an MPC tick or a planner call -- has a compute unit to itself and takes up to 150 KB: its
// trajectory mirrors (x, u, k, K, uMin, uMax of all t_cap rows: 70 KB for X = 7, U = 2, 300 rows) then sit in LDS
// too, instead of every stage of the dependent rollout and Riccati chains waiting on global memory.
// Priority: array parameters (every ctDynamics waits on them), then the trajectory mirrors.
// The rollouts may store their stage results straight to global memory when nothing in their loop LOADS from it (loads and
// stores of a wavefront share one counter and complete in order: a load behind a store waits for the store's acknowledgement --
// the reason the ring exists): the trajectory mirrors and every array parameter the model reads must be in LDS.
int roll_direct_ok(const tplhip_ilqr* h, const DevState& d)
{
    if (!d.lds_traj || tplhip::debug_opt("ilqr_no_direct")) return 0;
    int arrays = 0;
    for (int i = 0; i < TPL_N_PARAMS; ++i)
        if (g_params[i].is_array && h->parray[i].elems > 0 && h->parray[i].dev) ++arrays;
    return arrays == d.n_lds_arrays ? 1 : 0;
}

void plan_lds(tplhip_ilqr* h)
{
    DevState& d = h->dev;
    d.lds_traj = 0;
    d.roll_direct = 0;
    d.n_lds_arrays = 0;
    d.lds_dyn_flat = 0;
    h->lds_dyn_bytes = 0;
    if (!h->lds_aperture || tplhip::debug_opt("ilqr_no_lds")) return;
    const int per_cu = (h->batch + h->n_cu - 1) / std::max(1, h->n_cu);          // workgroups sharing a compute unit
    size_t wg_bytes = per_cu <= 1 ? (size_t)150 * 1024 : (per_cu == 2 ? (size_t)78 * 1024 : (size_t)39 * 1024);
    int kb = 0;
    if (tplhip::debug_opt("ilqr_lds_kb", &kb))                                   // tests: force the other home of the mirrors
        wg_bytes = std::min((size_t)150, std::max((size_t)8, (size_t)kb)) * 1024;
    const size_t budget = (wg_bytes - sizeof(Lds)) / sizeof(double);
    const size_t traj = (size_t)h->rows * (NX + NU + NU + NU * NX + NU + NU);
    size_t off = 0;
    if (traj <= budget) { d.lds_traj = 1; off = traj; }
    for (int i = 0; i < TPL_N_PARAMS; ++i) {
        if (!g_params[i].is_array) continue;
        const ArrayParam& a = h->parray[i];
        if (a.elems <= 0 || !a.dev) continue;
        if (off + (size_t)a.elems > budget) continue;
        const int k = d.n_lds_arrays++;
        d.lds_src[k] = a.dev;
        d.lds_src_stride[k] = a.per_problem ? a.elems : 0;
        d.lds_len[k] = (int)a.elems;
        d.lds_off[k] = (int)off;
        h->parray[i].lds_off = (long long)off;
        off += ((size_t)a.elems + 1) & ~(size_t)1;          // keep 16-byte alignment
    }
    h->lds_dyn_bytes = off * sizeof(double);
    d.lds_dyn_flat = h->lds_aperture + h->lds_static;
    d.roll_direct = roll_direct_ok(h, d);
}

// The split-launch kernels keep everything in dynamic LDS: the Lds block at the head, then what fits of the array
// parameters and the trajectory mirrors.  Their budget is what lets TPL_ILQR_CHAIN_WAVES workgroups per SIMD stay
// resident (160 KB per compute unit, four SIMDs): occupancy first, residency second -- co-resident wavefronts hide
// a wait on L2 that the single launch's lone wavefront would sit through.
void plan_lds_split(tplhip_ilqr* h)
{
    DevState& d = h->dev_split;
    d = h->dev;
    d.params = h->d_params_split;
    d.run = h->d_run;
    d.lds_traj = 0;
    d.roll_direct = 0;
    d.n_lds_arrays = 0;
    d.lds_dyn_flat = 0;
    d.lds_base_flat = h->lds_aperture;
    h->lds_dyn_bytes_split = (size_t)LDS_HEAD * sizeof(double);
    for (auto& a : h->parray) a.lds_off_split = -1;
    if (!h->lds_aperture || tplhip::debug_opt("ilqr_no_lds")) return;
    size_t wg_bytes = (size_t)160 * 1024 / (4 * TPL_ILQR_CHAIN_WAVES);
    int kb = 0;
    if (tplhip::debug_opt("ilqr_split_lds_kb", &kb)) wg_bytes = std::min((size_t)150, std::max((size_t)8, (size_t)kb)) * 1024;
    if (wg_bytes <= (size_t)LDS_HEAD * sizeof(double)) return;
    const size_t budget = wg_bytes / sizeof(double) - LDS_HEAD;
    const size_t traj = (size_t)h->rows * (NX + NU + NU + NU * NX + NU + NU);
    size_t arrays = 0;
    for (int i = 0; i < TPL_N_PARAMS; ++i)
        if (g_params[i].is_array && h->parray[i].elems > 0 && h->parray[i].dev) arrays += ((size_t)h->parray[i].elems + 1) & ~(size_t)1;
    size_t off = 0;
    if (traj + arrays <= budget) { d.lds_traj = 1; off = traj; }         // the mirrors only when the arrays fit as well
    for (int i = 0; i < TPL_N_PARAMS; ++i) {
        if (!g_params[i].is_array) continue;
        ArrayParam& a = h->parray[i];
        if (a.elems <= 0 || !a.dev) continue;
        if (off + (size_t)a.elems > budget) continue;
        const int k = d.n_lds_arrays++;
        d.lds_src[k] = a.dev;
        d.lds_src_stride[k] = a.per_problem ? a.elems : 0;
        d.lds_len[k] = (int)a.elems;
        d.lds_off[k] = (int)off;
        a.lds_off_split = (long long)off;
        off += ((size_t)a.elems + 1) & ~(size_t)1;
    }
    h->lds_dyn_bytes_split = (LDS_HEAD + off) * sizeof(double);
    d.lds_dyn_flat = h->lds_aperture + (unsigned long long)LDS_HEAD * sizeof(double);   // the split kernels have no static LDS
    // (the ring stays in the two-wavefront forms: with a second wavefront's stores in the same queue the direct form LOSES there,
    // 2 381 -> 2 345 batched it/s at 4 096, 1 233 -> 1 207 at 8 192, where the lone wavefront of the single launch gains 2 %;
    // profiles/r06_ilqr_direct_store_ab.txt)
    d.roll_direct = tplhip::debug_opt("ilqr_direct_split") ? roll_direct_ok(h, d) : 0;
}

// dev_split -> device memory (the split / fused kernels take a pointer), when it changed
int upload_dev_split(tplhip_ilqr* h)
{
    if (!h->d_dev_split || !h->h_dev_split) return TPLHIP_OK;
    if (h->dev_split_valid && std::memcmp(&h->dev_split_uploaded, &h->dev_split, sizeof(DevState)) == 0) return TPLHIP_OK;
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");        // the staging block may still be in flight
    std::memcpy(h->h_dev_split, &h->dev_split, sizeof(DevState));
    HIP_TRY(hipMemcpyAsync(h->d_dev_split, h->h_dev_split, sizeof(DevState), hipMemcpyHostToDevice, h->stream), "copying the solver state record failed");
    std::memcpy(&h->dev_split_uploaded, &h->dev_split, sizeof(DevState));
    h->dev_split_valid = true;
    return TPLHIP_OK;
}

int upload_params(tplhip_ilqr* h)
{
    // layout of the parameter arena: the Params records, then every non-empty array parameter
    auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    const size_t rec_bytes = pad(sizeof(Params) * (size_t)h->batch);      // one block of parameter records
    size_t need = 2 * rec_bytes;                                           // single-launch form, split-launch form
    for (int i = 0; i < TPL_N_PARAMS; ++i)
        if (g_params[i].is_array) need += pad(h->parray[i].host.size() * sizeof(double));
    if (need > h->parena_cap) {
        HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
        if (h->h_parena) (void)hipHostFree(h->h_parena);
        (void)hipFree(h->d_parena);
        h->h_parena = nullptr; h->d_parena = nullptr; h->parena_cap = 0;
        const size_t cap = need + need / 2;
        HIP_TRY(hipHostMalloc((void**)&h->h_parena, cap, hipHostMallocDefault), "parameter staging allocation failed");
        HIP_TRY(hipMalloc((void**)&h->d_parena, cap), "parameter array allocation failed");
        h->parena_cap = cap;
        h->d_params = reinterpret_cast<Params*>(h->d_parena);
        h->d_params_split = reinterpret_cast<Params*>(h->d_parena + rec_bytes);
        h->dev.params = h->d_params;
    }
    size_t off = 2 * rec_bytes;
    for (int i = 0; i < TPL_N_PARAMS; ++i) {
        if (!g_params[i].is_array) continue;
        ArrayParam& a = h->parray[i];
        const size_t bytes = a.host.size() * sizeof(double);
        a.dev = nullptr;
        if (!bytes) continue;
        std::memcpy(h->h_parena + off, a.host.data(), bytes);
        a.dev = reinterpret_cast<double*>(h->d_parena + off);
        off += pad(bytes);
    }
    for (auto& a : h->parray) a.lds_off = -1;
    plan_lds(h);
    plan_lds_split(h);
    if (int rc = upload_dev_split(h)) return rc;
    for (int form = 0; form < 2; ++form)
    for (int b = 0; b < h->batch; ++b) {
        Params& P = reinterpret_cast<Params*>(h->h_parena + (form ? rec_bytes : 0))[b];
        std::memset(&P, 0, sizeof(Params));
        for (int i = 0; i < TPL_N_PARAMS; ++i) {
            char* slot = (char*)&P + g_params[i].offset;
            if (g_params[i].is_array) {
                const ArrayParam& a = h->parray[i];
                DynArr d;
                d.data = a.dev ? a.dev + (a.per_problem ? (size_t)b * a.elems : 0) : nullptr;
                const long long lds_off = form ? a.lds_off_split : a.lds_off;
                const unsigned long long lds_base = form ? h->dev_split.lds_dyn_flat : h->dev.lds_dyn_flat;
                if (lds_off >= 0)       // the kernel copies the array to this LDS address first
                    d.data = (const double*)(uintptr_t)(lds_base + (unsigned long long)lds_off * sizeof(double));
                d.ndim = a.ndim;
                d.n0 = a.ndim > 0 ? a.dims[0] : 0;
                d.n1 = a.ndim > 1 ? a.dims[1] : 0;
                d.pad = 0;
                std::memcpy(slot, &d, sizeof d);
            } else {
                std::memcpy(slot, &h->pscalar[i][b], sizeof(double));
            }
        }
    }
    HIP_TRY(hipMemcpyAsync(h->d_parena, h->h_parena, off, hipMemcpyHostToDevice, h->stream), "copying parameters failed");
    return TPLHIP_OK;
}

void host_params(tplhip_ilqr* h, int b, Params* P)
{
    std::memset(P, 0, sizeof(Params));
    for (int i = 0; i < TPL_N_PARAMS; ++i) {
        char* slot = (char*)P + g_params[i].offset;
        if (g_params[i].is_array) {
            const ArrayParam& a = h->parray[i];
            DynArr d;
            d.data = a.host.empty() ? nullptr : a.host.data() + (a.per_problem ? (size_t)b * a.elems : 0);
            d.ndim = a.ndim;
            d.n0 = a.ndim > 0 ? a.dims[0] : 0;
            d.n1 = a.ndim > 1 ? a.dims[1] : 0;
            d.pad = 0;
            std::memcpy(slot, &d, sizeof d);
        } else {
            std::memcpy(slot, &h->pscalar[i][b], sizeof(double));
        }
    }
}

int upload_inputs(tplhip_ilqr* h)
{
    HIP_TRY(hipMemcpyAsync(h->d_arena + h->in_lo, h->h_arena + h->in_lo, (h->in_hi - h->in_lo) * sizeof(double),
                           hipMemcpyHostToDevice, h->stream), "copying solver state failed");
    for (int f : kInputFields) h->device_newer[f] = false;
    return upload_params(h);
}

// Which form runs.  A two-wavefront form (split sequence or fused launch) needs one integrator and one order for the whole
// batch (they are template parameters of the chain phase) and pays off when the batch holds more problems than the single
// launch keeps resident AND that launch is held to one wavefront per SIMD by its registers.  Measured for all seven tpl
// problems at 1 536 ... 8 192 problems (tools/ab_ilqr_forms_all.py, profiles/r04_ilqr_split_ab.txt; fused against single
// launch): five of them hold four wavefronts per compute unit in the single launch and gain -- the two tracking MPCs
// +28 ... +38 %, lateral_profile (T = 120) +22 ... +34 %, velocity_profile_space +31 ... +42 %, velocity_profile_time
// +27 ... +37 % -- and the two reference-line smoothers, whose single launch already holds eight, are within 1-3 % either
// way.  So every model takes the fused form above 4 x CUs problems.  (Asking the runtime for k_ilqr_update's residency
// -- hipOccupancyMaxActiveBlocksPerMultiprocessor -- was tried as the criterion and dropped: in a process where another
// handle had raised the kernel's dynamic-LDS allowance it answers four for the smoothers instead of eight, or refuses with
// hipErrorUnknown.  The split SEQUENCE is the A/B leg only: light models do not pay for its launches.)
//
// 0 = k_ilqr_update (one launch), 1 = the split sequence, 2 = k_ilqr_fused (one launch, the phases as calls, two
// wavefronts per SIMD).  TPLHIP_DEBUG ilqr_split=0|1|2 forces one (tests run all three).
int pick_form(const tplhip_ilqr* h)
{
    if (!h->split_uniform || !h->d_run || !h->d_dev_split || !h->lds_aperture) return 0;
    if (h->split_mode >= 0) return h->split_mode;
    return (long long)h->batch > 4LL * h->n_cu ? 2 : 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a setting per FUNCTION and DEVICE, shared by every handle of this model
// library: a process-global table keeps the largest value set so far and only ever raises it (a second handle with a
// smaller requirement must not lower the limit under the first one's feet).
int raise_lds_limit(int device, const void* fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> limit;
    if (bytes <= 48 * 1024) return TPLHIP_OK;                 // what every launch gets without asking
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = limit[{device, fn}];
    if (bytes > cur) {
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "raising the LDS limit failed");
        cur = bytes;
    }
    return TPLHIP_OK;
}

template <bool SO, int INT>
int launch_chain(tplhip_ilqr* h, int lg, int it)
{
    const void* fn = reinterpret_cast<const void*>(&k_split_chain<SO, INT>);
    // more dynamic LDS than a launch gets without asking (only with TPLHIP_DEBUG=ilqr_split_lds_kb beyond 48): opt in, per
    // instantiation -- a batch may come back with another integrator, i.e. another kernel
    if (int rc = raise_lds_limit(h->device, fn, h->lds_dyn_bytes_split)) return rc;
    hipLaunchKernelGGL((k_split_chain<SO, INT>), dim3(h->batch), dim3(WAVE), h->lds_dyn_bytes_split, h->stream, h->d_dev_split, lg, it);
    return TPLHIP_OK;
}

int launch_split(tplhip_ilqr* h)
{
    const size_t lds = h->lds_dyn_bytes_split;
    if (int rc = raise_lds_limit(h->device, reinterpret_cast<const void*>(k_split_begin), lds)) return rc;
    if (int rc = raise_lds_limit(h->device, reinterpret_cast<const void*>(k_split_refresh), lds)) return rc;
    const int rounds = std::max(1, h->max_lg_max);
    for (int lg = 0; lg < rounds; ++lg) {
        hipLaunchKernelGGL(k_split_begin, dim3(h->batch), dim3(WAVE), lds, h->stream, h->d_dev_split, lg);
        if (lg >= h->max_lg_max) break;
        for (int it = 0; it < h->max_iter_max; ++it) {
            hipLaunchKernelGGL(k_split_refresh, dim3(h->batch, (h->horizon_max + WAVE - 1) / WAVE), dim3(WAVE), lds, h->stream, h->d_dev_split, lg, it);
            int rc;
            const int ty = h->split_int_type;
            if (h->split_second_order) rc = ty == EULER ? launch_chain<true, EULER>(h, lg, it) : (ty == HEUN ? launch_chain<true, HEUN>(h, lg, it) : launch_chain<true, RK4>(h, lg, it));
            else rc = ty == EULER ? launch_chain<false, EULER>(h, lg, it) : (ty == HEUN ? launch_chain<false, HEUN>(h, lg, it) : launch_chain<false, RK4>(h, lg, it));
            if (rc) return rc;
        }
    }
    hipLaunchKernelGGL(k_split_end, dim3((h->batch + 255) / 256), dim3(256), 0, h->stream, h->d_dev_split, h->batch);
    return TPLHIP_OK;
}

template <bool SO, int INT, bool ONE>
int launch_fused_t(tplhip_ilqr* h)
{
    const void* fn = ONE ? reinterpret_cast<const void*>(&k_ilqr_fused1<SO, INT>) : reinterpret_cast<const void*>(&k_ilqr_fused<SO, INT>);
    if (int rc = raise_lds_limit(h->device, fn, h->lds_dyn_bytes_split)) return rc;
    if (ONE)
        hipLaunchKernelGGL((k_ilqr_fused1<SO, INT>), dim3(h->batch), dim3(WAVE), h->lds_dyn_bytes_split, h->stream, h->d_dev_split,
                           std::max(1, h->max_lg_max), h->max_iter_max, (h->horizon_max + WAVE - 1) / WAVE);
    else
        hipLaunchKernelGGL((k_ilqr_fused<SO, INT>), dim3(h->batch), dim3(WAVE), h->lds_dyn_bytes_split, h->stream, h->d_dev_split,
                           std::max(1, h->max_lg_max), h->max_iter_max, (h->horizon_max + WAVE - 1) / WAVE);
    return TPLHIP_OK;
}

template <bool ONE>
int launch_fused(tplhip_ilqr* h)
{
    const int ty = h->split_int_type;
    if (h->split_second_order) return ty == EULER ? launch_fused_t<true, EULER, ONE>(h) : (ty == HEUN ? launch_fused_t<true, HEUN, ONE>(h) : launch_fused_t<true, RK4, ONE>(h));
    return ty == EULER ? launch_fused_t<false, EULER, ONE>(h) : (ty == HEUN ? launch_fused_t<false, HEUN, ONE>(h) : launch_fused_t<false, RK4, ONE>(h));
}

int launch(tplhip_ilqr* h)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        HIP_TRY(hipEventCreate(&e0), "event creation failed");
        HIP_TRY(hipEventCreate(&e1), "event creation failed");
        HIP_TRY(hipEventRecord(e0, h->stream), "event record failed");
    }
    const int form = pick_form(h);
    if (form == 3) {
        h->last_form = 3;
        if (int rc = launch_fused<true>(h)) return rc;
    } else if (form == 2) {
        h->last_form = 2;
        if (int rc = launch_fused<false>(h)) return rc;
    } else if (form == 1) {
        h->last_form = 1;
        if (int rc = launch_split(h)) return rc;
    } else {
        h->last_form = 0;
        if (int rc = raise_lds_limit(h->device, reinterpret_cast<const void*>(k_ilqr_update), h->lds_dyn_bytes)) return rc;      // more than the default allowance: opt in
        hipLaunchKernelGGL(k_ilqr_update, dim3(h->batch), dim3(WAVE), h->lds_dyn_bytes, h->stream, h->dev);
    }
    HIP_TRY(hipGetLastError(), "solver launch failed");
    if (h->timing) {
        HIP_TRY(hipEventRecord(e1, h->stream), "event record failed");
        h->events.emplace_back(e0, e1);
        h->launches += 1;
    }
    for (int f = 0; f < TPL_F_COUNT; ++f) {
        const bool input_only = f == TPL_F_BARRIER_WEIGHT || f == TPL_F_LG_MULT_LIMIT || f == TPL_F_U_MAX
                                || f == TPL_F_U_MIN || f == TPL_F_PREV_U;
        if (!input_only) h->device_newer[f] = true;
    }
    return TPLHIP_OK;
}

// the scalars and all of kEagerFields in one copy (they are one range of the arena)
int fetch_eager(tplhip_ilqr* h)
{
    HIP_TRY(hipMemcpyAsync(h->h_arena + h->out_lo, h->d_arena + h->out_lo, (h->out_hi - h->out_lo) * sizeof(double),
                           hipMemcpyDeviceToHost, h->stream), "copying solver results failed");
    for (int f : kEagerFields) h->device_newer[f] = false;
    return TPLHIP_OK;
}

int fetch_field(tplhip_ilqr* h, int f)
{
    if (!h->device_newer[f]) return TPLHIP_OK;
    HIP_TRY(hipMemcpyAsync(h->host[f].data(), h->dev.f[f], field_elems(h, f) * sizeof(double),
                           hipMemcpyDeviceToHost, h->stream), "copying solver results failed");
    h->device_newer[f] = false;
    return TPLHIP_OK;
}

// The reference template asserts the shapes of array parameters inside its builtins
// (check(ARR.ndim == 1) ..., optim.c:375-486) and surfaces the first failure as an
// AssertionError after update() (:1195-1200).  Same conditions, checked before the launch.
int check_param_shapes(tplhip_ilqr* h)
{
    static const int nd[][2] = TPL_NDIM_CHECKS;
    static const int same[][2] = TPL_SAME_LEN_CHECKS;
    for (int k = 0; nd[k][0] >= 0; ++k) {
        const ArrayParam& a = h->parray[nd[k][0]];
        if (a.ndim != nd[k][1])
            return fail(TPLHIP_EASSERT, "params.%s.ndim == %d", g_params[nd[k][0]].name, nd[k][1]);
    }
    for (int k = 0; same[k][0] >= 0; ++k) {
        const ArrayParam& a = h->parray[same[k][0]];
        const ArrayParam& b = h->parray[same[k][1]];
        if (a.dims[0] != b.dims[0])
            return fail(TPLHIP_EASSERT, "params.%s.dims[0] == params.%s.dims[0]", g_params[same[k][1]].name, g_params[same[k][0]].name);
    }
    return TPLHIP_OK;
}

int check_settings(tplhip_ilqr* h)
{
    if (int rc = check_param_shapes(h)) return rc;
    for (int b = 0; b < h->batch; ++b) {
        double* sc = h->scalars.data() + (size_t)b * TPL_S_COUNT;
        const int T = (int)sc[TPL_S_HORIZON];
        if (T < 1 || T > h->t_cap)
            return fail(TPLHIP_EINVAL, "horizon %d outside [1, %d] (t_cap of this solver object)", T, h->t_cap);
        const int os = (int)sc[TPL_S_OPT_START];
        if (os < 0 || os > T) return fail(TPLHIP_EINVAL, "opt_start %d outside [0, horizon]", os);
        const int it = (int)sc[TPL_S_INTEGRATOR_TYPE];
        if (it < 0 || it > 2) return fail(TPLHIP_EINVAL, "unknown integrator type %d", it);
        // what the split-launch form needs to know about the batch as a whole
        const bool so = sc[TPL_S_USE_QUADRATIC_TERMS] != 0.0;
        const int mi = (int)sc[TPL_S_MAX_ITERATIONS], ml = (int)sc[TPL_S_MAX_LG_ITERATIONS];
        if (b == 0) { h->split_uniform = true; h->split_int_type = it; h->split_second_order = so; h->max_iter_max = mi; h->max_lg_max = ml; h->horizon_max = T; }
        else {
            if (it != h->split_int_type || so != h->split_second_order) h->split_uniform = false;
            h->max_iter_max = std::max(h->max_iter_max, mi);
            h->max_lg_max = std::max(h->max_lg_max, ml);
            h->horizon_max = std::max(h->horizon_max, T);
        }
    }
    return TPLHIP_OK;
}

} // namespace

extern "C" {

const char* tplhip_ilqr_model_name(void) { return TPL_MODEL_NAME; }
void tplhip_ilqr_dims(int* X, int* U, int* C, int* n_params)
{
    if (X) *X = NX;
    if (U) *U = NU;
    if (C) *C = NC;
    if (n_params) *n_params = TPL_N_PARAMS;
}
const char* tplhip_ilqr_param_name(int i) { return (i >= 0 && i < TPL_N_PARAMS) ? g_params[i].name : nullptr; }
int tplhip_ilqr_param_is_array(int i) { return (i >= 0 && i < TPL_N_PARAMS) ? g_params[i].is_array : -1; }
const char* tplhip_ilqr_last_error(void) { return g_error; }
const char* tplhip_ilqr_fingerprint(void) { return TPL_MODEL_FINGERPRINT; }

int tplhip_ilqr_create(int device, int batch, int t_cap, tplhip_ilqr** out)
{
    if (!out) return fail(TPLHIP_EINVAL, "null out pointer");
    if (batch < 1) return fail(TPLHIP_EINVAL, "batch must be >= 1");
    if (t_cap < 1 || t_cap > TPLHIP_ILQR_H_MAX - 1)
        return fail(TPLHIP_EINVAL, "t_cap %d outside [1, %d]", t_cap, TPLHIP_ILQR_H_MAX - 1);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(TPLHIP_EDEVICE, "no usable HIP device (requested %d)", device);
    if (device < 0) device = n - 1;
    if (device >= n) return fail(TPLHIP_EDEVICE, "no usable HIP device (requested %d)", device);
    HIP_TRY(hipSetDevice(device), "hipSetDevice failed");

    tplhip_ilqr* h = new (std::nothrow) tplhip_ilqr();
    if (!h) return fail(TPLHIP_EDEVICE, "out of host memory");
    h->device = device;
    h->batch = batch;
    h->t_cap = t_cap;
    h->rows = t_cap + 1;
    h->dev.rows = h->rows;
    hipError_t err = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);      // nothing of a solver runs on the null stream
    if (err != hipSuccess) { delete h; return fail(TPLHIP_EDEVICE, "%s: stream creation failed", hipGetErrorString(err)); }

    {
        // arena layout: kArenaOrder first, then the fields update() only fetches on demand
        std::vector<int> order(std::begin(kArenaOrder), std::end(kArenaOrder));
        for (int f = 0; f < TPL_F_COUNT; ++f)
            if (std::find(order.begin(), order.end(), f) == order.end()) order.push_back(f);
        auto pad = [](size_t n) { return (n + kArenaAlign - 1) / kArenaAlign * kArenaAlign; };
        size_t off[TPL_F_COUNT + 1] = {0}, off_scalars = 0, total = 0;
        for (int f : order) {
            (f < 0 ? off_scalars : off[f]) = total;
            total += pad(f < 0 ? (size_t)batch * TPL_S_COUNT : field_elems(h, f));
        }
        h->arena_elems = total;
        err = hipHostMalloc((void**)&h->h_arena, total * sizeof(double), hipHostMallocDefault);
        if (err == hipSuccess) err = hipMalloc((void**)&h->d_arena, total * sizeof(double));
        if (err == hipSuccess) err = hipMemsetAsync(h->d_arena, 0, total * sizeof(double), h->stream);
        if (err != hipSuccess) { tplhip_ilqr_destroy(h); return fail(TPLHIP_EDEVICE, "%s: solver state allocation failed", hipGetErrorString(err)); }
        std::memset(h->h_arena, 0, total * sizeof(double));
        for (int f = 0; f < TPL_F_COUNT; ++f) {
            h->host[f].p = h->h_arena + off[f];
            h->host[f].n = field_elems(h, f);
            h->dev.f[f] = h->d_arena + off[f];
        }
        h->scalars.p = h->h_arena + off_scalars;
        h->scalars.n = (size_t)batch * TPL_S_COUNT;
        h->dev.scalars = h->d_arena + off_scalars;
        h->in_lo = off[kArenaOrder[0]];
        h->in_hi = off[TPL_F_LAGRANGE_MULTIPLIER] + pad(field_elems(h, TPL_F_LAGRANGE_MULTIPLIER));
        h->out_lo = off_scalars;
        h->out_hi = off[TPL_F_G] + pad(field_elems(h, TPL_F_G));
    }
    h->pscalar.assign(TPL_N_PARAMS, std::vector<double>());
    h->parray.assign(TPL_N_PARAMS, ArrayParam());
    for (int i = 0; i < TPL_N_PARAMS; ++i)
        if (!g_params[i].is_array) h->pscalar[i].assign(batch, 0.0);
    h->hparams.resize(batch);
    const size_t cand_x = (size_t)batch * N_ALPHA * h->rows * NX, cand_u = (size_t)batch * N_ALPHA * h->rows * NU;
    err = hipMalloc(&h->dev.cand_x, cand_x * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&h->dev.cand_u, cand_u * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&h->dev.cand_c, (size_t)batch * N_ALPHA * h->rows * sizeof(double));
    if (err == hipSuccess) err = hipMalloc((void**)&h->d_run, (size_t)batch * sizeof(int));
    if (err == hipSuccess) err = hipMalloc((void**)&h->d_dev_split, sizeof(DevState));
    if (err == hipSuccess) err = hipHostMalloc((void**)&h->h_dev_split, sizeof(DevState), hipHostMallocDefault);
    if (err == hipSuccess) err = hipMemsetAsync(h->d_run, 0, (size_t)batch * sizeof(int), h->stream);
    h->dev.run = h->d_run;
    {
        int v = 0;
        if (tplhip::debug_opt("ilqr_split", &v)) h->split_mode = (v >= 0 && v <= 3) ? v : 1;
        // the split kernels place their Lds block at dynamic-LDS offset 0 and the host computes LDS addresses from that:
        // it only holds while none of them has static LDS of its own (none has; a compiler that adds some turns the form off)
        hipFuncAttributes fa;
        const void* fns[] = {reinterpret_cast<const void*>(k_split_begin), reinterpret_cast<const void*>(k_split_refresh),
                             reinterpret_cast<const void*>(&k_split_chain<true, EULER>), reinterpret_cast<const void*>(&k_split_chain<true, HEUN>),
                             reinterpret_cast<const void*>(&k_split_chain<true, RK4>), reinterpret_cast<const void*>(&k_split_chain<false, EULER>),
                             reinterpret_cast<const void*>(&k_split_chain<false, HEUN>), reinterpret_cast<const void*>(&k_split_chain<false, RK4>),
                             reinterpret_cast<const void*>(&k_ilqr_fused<true, EULER>), reinterpret_cast<const void*>(&k_ilqr_fused<true, HEUN>),
                             reinterpret_cast<const void*>(&k_ilqr_fused<true, RK4>), reinterpret_cast<const void*>(&k_ilqr_fused<false, EULER>),
                             reinterpret_cast<const void*>(&k_ilqr_fused<false, HEUN>), reinterpret_cast<const void*>(&k_ilqr_fused<false, RK4>),
                             reinterpret_cast<const void*>(&k_ilqr_fused1<true, EULER>), reinterpret_cast<const void*>(&k_ilqr_fused1<true, HEUN>),
                             reinterpret_cast<const void*>(&k_ilqr_fused1<true, RK4>), reinterpret_cast<const void*>(&k_ilqr_fused1<false, EULER>),
                             reinterpret_cast<const void*>(&k_ilqr_fused1<false, HEUN>), reinterpret_cast<const void*>(&k_ilqr_fused1<false, RK4>)};
        for (const void* fn : fns)
            if (hipFuncGetAttributes(&fa, fn) != hipSuccess || fa.sharedSizeBytes != 0) h->split_mode = 0;
        (void)hipGetLastError();
    }
    if (err != hipSuccess) { tplhip_ilqr_destroy(h); return fail(TPLHIP_EDEVICE, "%s: solver workspace allocation failed", hipGetErrorString(err)); }
    h->dev.params = nullptr;                                // set by the first upload_params
    h->dev.stamps = nullptr;
    {
        // where LDS sits in the flat address space (constant per process and device)
        unsigned long long* d_ap = nullptr;
        hipFuncAttributes attr;
        if (hipMalloc(&d_ap, sizeof(unsigned long long)) == hipSuccess) {
            hipLaunchKernelGGL(k_lds_aperture, dim3(1), dim3(1), 16, h->stream, d_ap);
            unsigned long long ap = 0;
            if (hipMemcpyAsync(&ap, d_ap, sizeof ap, hipMemcpyDeviceToHost, h->stream) == hipSuccess
                && hipStreamSynchronize(h->stream) == hipSuccess
                && hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_ilqr_update)) == hipSuccess) {
                h->lds_aperture = ap;
                h->lds_static = attr.sharedSizeBytes;
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0)
                    h->n_cu = prop.multiProcessorCount;
            }
            (void)hipFree(d_ap);
        }
    }
#ifdef TPL_ILQR_STAMP
    if (hipMalloc(&h->dev.stamps, sizeof(long long) * 16 * (size_t)batch) != hipSuccess) h->dev.stamps = nullptr;
#endif

    // defaults of pyOptimAlloc, optim.c:1894-1921
    const int T0 = std::min(20, t_cap);
    for (int b = 0; b < batch; ++b) {
        double* sc = h->scalars.data() + (size_t)b * TPL_S_COUNT;
        sc[TPL_S_DT] = 0.05;
        sc[TPL_S_HORIZON] = T0;
        sc[TPL_S_MIN_REL_COST_CHANGE] = 1e-6;
        sc[TPL_S_MAX_ITERATIONS] = 5;
        sc[TPL_S_MAX_LG_ITERATIONS] = 1;
        sc[TPL_S_USE_QUADRATIC_TERMS] = 1.0;
        for (int c = 0; c < NCM; ++c) {
            h->host[TPL_F_LG_MULT_LIMIT][(size_t)b * NCM + c] = INFINITY;
            h->host[TPL_F_BARRIER_WEIGHT][(size_t)b * NCM + c] = 1.0;
        }
        for (int t = 0; t < T0; ++t)
            for (int d = 0; d < NU; ++d) {
                h->host[TPL_F_U_MAX][((size_t)b * h->rows + t) * NU + d] = INFINITY;
                h->host[TPL_F_U_MIN][((size_t)b * h->rows + t) * NU + d] = -INFINITY;
            }
    }
    *out = h;
    return TPLHIP_OK;
}

int tplhip_ilqr_destroy(tplhip_ilqr* h)
{
    if (!h) return TPLHIP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& ev : h->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipFree(h->d_arena);
    if (h->h_arena) (void)hipHostFree(h->h_arena);
    (void)hipFree(h->d_parena);
    if (h->h_parena) (void)hipHostFree(h->h_parena);
    (void)hipFree(h->dev.cand_x);
    (void)hipFree(h->dev.cand_u);
    (void)hipFree(h->dev.cand_c);
    (void)hipFree(h->d_run);
    (void)hipFree(h->d_dev_split);
    if (h->h_dev_split) (void)hipHostFree(h->h_dev_split);
    for (double* p : h->d_snapshot) (void)hipFree(p);
    (void)hipFree(h->d_snapshot_scalars);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return TPLHIP_OK;
}

int tplhip_ilqr_batch(tplhip_ilqr* h) { return h ? h->batch : 0; }
int tplhip_ilqr_t_cap(tplhip_ilqr* h) { return h ? h->t_cap : 0; }

double* tplhip_ilqr_ptr(tplhip_ilqr* h, int field, int64_t* problem_stride)
{
    if (!h || field < 0 || field >= TPL_F_COUNT) { fail(TPLHIP_EINVAL, "bad field id %d", field); return nullptr; }
    if (h->device_newer[field]) {
        (void)hipSetDevice(h->device);
        if (fetch_field(h, field) != TPLHIP_OK) return nullptr;
        if (hipStreamSynchronize(h->stream) != hipSuccess) { fail(TPLHIP_EDEVICE, "copying solver results failed"); return nullptr; }
    }
    if (problem_stride) *problem_stride = (int64_t)((field_is_per_stage(field) ? h->rows : 1) * field_width(field));
    return h->host[field].data();
}

double* tplhip_ilqr_scalars(tplhip_ilqr* h) { return h ? h->scalars.data() : nullptr; }

double* tplhip_ilqr_param_scalar_ptr(tplhip_ilqr* h, int i)
{
    if (!h || i < 0 || i >= TPL_N_PARAMS || g_params[i].is_array) return nullptr;
    return h->pscalar[i].data();
}

int tplhip_ilqr_set_param_array(tplhip_ilqr* h, int i, const double* data, int ndim, const int64_t* dims, int per_problem)
{
    if (!h || i < 0 || i >= TPL_N_PARAMS || !g_params[i].is_array) return fail(TPLHIP_EINVAL, "parameter %d is not an array parameter", i);
    if (ndim < 0 || ndim > 8) return fail(TPLHIP_EINVAL, "bad ndim %d", ndim);
    ArrayParam& a = h->parray[i];
    int64_t elems = ndim > 0 ? 1 : 0;
    for (int d = 0; d < ndim; ++d) { a.dims[d] = dims[d]; elems *= dims[d]; }
    a.ndim = ndim;
    a.elems = elems;
    a.per_problem = per_problem ? 1 : 0;
    const size_t total = (size_t)elems * (per_problem ? h->batch : 1);
    if (total && !data) return fail(TPLHIP_EINVAL, "null data");
    a.host.assign(data, data + total);
    return TPLHIP_OK;
}

int tplhip_ilqr_get_param_array(tplhip_ilqr* h, int i, const double** data, int* ndim, int64_t* dims, int* per_problem)
{
    if (!h || i < 0 || i >= TPL_N_PARAMS || !g_params[i].is_array) return fail(TPLHIP_EINVAL, "parameter %d is not an array parameter", i);
    const ArrayParam& a = h->parray[i];
    if (data) *data = a.host.data();
    if (ndim) *ndim = a.ndim;
    if (dims) for (int d = 0; d < a.ndim; ++d) dims[d] = a.dims[d];
    if (per_problem) *per_problem = a.per_problem;
    return TPLHIP_OK;
}

int tplhip_ilqr_update(tplhip_ilqr* h)
{
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    int rc = check_settings(h);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const auto t0 = std::chrono::steady_clock::now();
    rc = upload_inputs(h);
    if (rc) return rc;
    rc = launch(h);
    if (rc) return rc;
    rc = fetch_eager(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream), "solver update failed");
    for (int b = 0; b < h->batch; ++b)
        if (h->scalars[(size_t)b * TPL_S_COUNT + TPL_S_TERMINATION_CONDITION] == -99.0)
            return fail(TPLHIP_EDEVICE, "LDS aperture mismatch: the solver kernel did not find its LDS where the host placed the parameter arrays");
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int b = 0; b < h->batch; ++b) h->scalars[(size_t)b * TPL_S_COUNT + TPL_S_RUNTIME] = ms;   // optim.c:1153-1159
    return TPLHIP_OK;
}

int tplhip_ilqr_shift(tplhip_ilqr* h, int amount)
{
    // shift(), optim.c:1162-1177, on the host mirrors
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    if (amount < 0) amount = 0;
    for (int f : {TPL_F_X, TPL_F_U, TPL_F_LAGRANGE_MULTIPLIER}) {
        int64_t st = 0;
        if (!tplhip_ilqr_ptr(h, f, &st)) return TPLHIP_EDEVICE;      // make sure the mirror is current
    }
    for (int b = 0; b < h->batch; ++b) {
        const double* sc = h->scalars.data() + (size_t)b * TPL_S_COUNT;
        const int T = (int)sc[TPL_S_HORIZON], os = (int)sc[TPL_S_OPT_START];
        double* x = h->host[TPL_F_X].data() + (size_t)b * h->rows * NX;
        double* u = h->host[TPL_F_U].data() + (size_t)b * h->rows * NU;
        double* lm = h->host[TPL_F_LAGRANGE_MULTIPLIER].data() + (size_t)b * h->rows * NCM;
        for (int t = os; t < T + 1; ++t)
            std::memmove(x + (size_t)t * NX, x + (size_t)std::min(T, t + amount) * NX, sizeof(double) * NX);
        for (int t = os; t < T; ++t)
            std::memmove(u + (size_t)t * NU, u + (size_t)std::min(T - 1, t + amount) * NU, sizeof(double) * NU);
        for (int t = os; t < T; ++t)
            std::memmove(lm + (size_t)t * NCM, lm + (size_t)std::min(T - 1, t + amount) * NCM, sizeof(double) * NCM);
    }
    return TPLHIP_OK;
}

static void host_dynamics(const Params& P, const double* x, const double* u, int t, double dt, int int_type, double* res)
{
    // optim.c:657-730 on the host, for Optim.dynamics (dead-time compensation in the MPC controllers)
    double k1[NX], k2[NX], k3[NX], k4[NX], tmp[NX];
    ctDynamics(P, x, u, (double)t, dt, k1, 1);
    if (int_type == EULER) {
        for (int i = 0; i < NX; ++i) res[i] = x[i] + k1[i] * dt;
    } else if (int_type == HEUN) {
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k1[i] * dt;
        ctDynamics(P, tmp, u, (double)t, dt, k2, 1);
        for (int i = 0; i < NX; ++i) res[i] = x[i] + (k1[i] + k2[i]) * (dt / 2.0);
    } else {
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k1[i] * (dt / 2.0);
        ctDynamics(P, tmp, u, (double)t, dt, k2, 1);
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k2[i] * (dt / 2.0);
        ctDynamics(P, tmp, u, (double)t, dt, k3, 1);
        for (int i = 0; i < NX; ++i) tmp[i] = x[i] + k3[i] * dt;
        ctDynamics(P, tmp, u, (double)t, dt, k4, 1);
        for (int i = 0; i < NX; ++i) {
            double acc = k1[i] + 0.0;
            acc = k2[i] * 2.0 + acc;
            acc = k3[i] * 2.0 + acc;
            acc = k4[i] + acc;
            res[i] = x[i] + acc * (dt / 6.0);
        }
    }
}

int tplhip_ilqr_dynamics(tplhip_ilqr* h, int b, const double* x, const double* u, int t, double dt, double* out)
{
    if (!h || !x || !u || !out || b < 0 || b >= h->batch) return fail(TPLHIP_EINVAL, "bad argument");
    Params P;
    host_params(h, b, &P);
    const int int_type = (int)h->scalars[(size_t)b * TPL_S_COUNT + TPL_S_INTEGRATOR_TYPE];
    host_dynamics(P, x, u, t, dt, int_type, out);
    return TPLHIP_OK;
}

int tplhip_ilqr_ct_dynamics(tplhip_ilqr* h, int b, const double* x, const double* u, int t, double dt, double* out)
{
    if (!h || !x || !u || !out || b < 0 || b >= h->batch) return fail(TPLHIP_EINVAL, "bad argument");
    Params P;
    host_params(h, b, &P);
    ctDynamics(P, x, u, (double)t, dt, out, 1);
    return TPLHIP_OK;
}

int tplhip_ilqr_upload(tplhip_ilqr* h)
{
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    int rc = check_settings(h);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    rc = upload_inputs(h);
    if (rc) return rc;
    const int snap_fields[3] = {TPL_F_X, TPL_F_U, TPL_F_LAGRANGE_MULTIPLIER};
    for (int i = 0; i < 3; ++i) {
        const size_t bytes = field_elems(h, snap_fields[i]) * sizeof(double);
        if (!h->d_snapshot[i]) HIP_TRY(hipMalloc(&h->d_snapshot[i], bytes), "snapshot allocation failed");
        HIP_TRY(hipMemcpyAsync(h->d_snapshot[i], h->dev.f[snap_fields[i]], bytes, hipMemcpyDeviceToDevice, h->stream), "snapshot copy failed");
    }
    if (!h->d_snapshot_scalars) HIP_TRY(hipMalloc(&h->d_snapshot_scalars, h->scalars.size() * sizeof(double)), "snapshot allocation failed");
    HIP_TRY(hipMemcpyAsync(h->d_snapshot_scalars, h->dev.scalars, h->scalars.size() * sizeof(double), hipMemcpyDeviceToDevice, h->stream), "snapshot copy failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "upload failed");
    h->have_snapshot = true;
    return TPLHIP_OK;
}

int tplhip_ilqr_restore(tplhip_ilqr* h)
{
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    if (!h->have_snapshot) return fail(TPLHIP_ESTATE, "nothing uploaded");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const int snap_fields[3] = {TPL_F_X, TPL_F_U, TPL_F_LAGRANGE_MULTIPLIER};
    for (int i = 0; i < 3; ++i)
        HIP_TRY(hipMemcpyAsync(h->dev.f[snap_fields[i]], h->d_snapshot[i], field_elems(h, snap_fields[i]) * sizeof(double),
                               hipMemcpyDeviceToDevice, h->stream), "snapshot restore failed");
    HIP_TRY(hipMemcpyAsync(h->dev.scalars, h->d_snapshot_scalars, h->scalars.size() * sizeof(double),
                           hipMemcpyDeviceToDevice, h->stream), "snapshot restore failed");
    return TPLHIP_OK;
}

int tplhip_ilqr_run_resident(tplhip_ilqr* h)
{
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    if (!h->have_snapshot) return fail(TPLHIP_ESTATE, "nothing uploaded");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    return launch(h);
}

int tplhip_ilqr_sync(tplhip_ilqr* h)
{
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "solver update failed");
    return TPLHIP_OK;
}

int tplhip_ilqr_download(tplhip_ilqr* h)
{
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
#ifdef TPL_ILQR_STAMP
    if (h->dev.stamps) {
        std::vector<long long> st((size_t)h->batch * 16);
        hipStreamSynchronize(h->stream);
        hipMemcpy(st.data(), h->dev.stamps, st.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double sum[16] = {0};
        for (int b = 0; b < h->batch; ++b) for (int i = 0; i < 16; ++i) sum[i] += (double)st[(size_t)b * 16 + i];
        const char* names[16] = {"initial rollout", "multiplier update", "derivatives", "backward pass (rest)", "ls: ring flush", "accept/copy",
                                 "bw: regs->LDS + fetch", "bw: phase A", "bw: phase B", "bw: phase C", "bw: phase D", "ls: dynamics chains (rest)", "ls: cost pass", "ls: sums", "ls/step: feedback law u", "ls/step: dynamics + stores"};
        for (int i = 0; i < 16; ++i) std::fprintf(stderr, "[stamp] %-22s %12.0f s_memtime ticks per problem per update\n", names[i], sum[i] / h->batch);
    }
#endif
    int rc = fetch_eager(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream), "solver update failed");
    return TPLHIP_OK;
}

#ifndef TPL_ILQR_SCHED
#define TPL_ILQR_SCHED "default"
#endif
// the same label as a marker in the file itself: genopt.compile_model reads it (no GPU, no dlopen) before it reuses a cached library
__attribute__((used)) static const char k_sched_marker[] = "tplhip_ilqr_sched=" TPL_ILQR_SCHED;

int tplhip_ilqr_info(tplhip_ilqr* h, char* buf, int n)
{
    if (!h || !buf || n < 1) return fail(TPLHIP_EINVAL, "bad argument");
    const bool sp = h->last_form != 0;
    // TPL_ILQR_SCHED: the instruction scheduler strategy this library was compiled under (genopt.compile_model passes it)
#ifndef TPL_ILQR_SCHED
#define TPL_ILQR_SCHED "default"
#endif
    std::snprintf(buf, (size_t)n, "{\"batch\": %d, \"t_cap\": %d, \"lds_traj\": %d, \"lds_arrays\": %d, \"lds_dyn_bytes\": %zu, "
                  "\"lds_static_bytes\": %zu, \"form\": \"%s\", \"chain_waves_per_simd\": %d, \"scheduler\": \"%s\"}", h->batch, h->rows - 1,
                  sp ? h->dev_split.lds_traj : h->dev.lds_traj, sp ? h->dev_split.n_lds_arrays : h->dev.n_lds_arrays,
                  sp ? h->lds_dyn_bytes_split : h->lds_dyn_bytes, sp ? (size_t)0 : h->lds_static,
                  h->last_form == 3 ? "fused1" : (h->last_form == 2 ? "fused" : (sp ? "split" : "single")), (sp && h->last_form != 3) ? TPL_ILQR_CHAIN_WAVES : 1,
                  TPL_ILQR_SCHED);
    return TPLHIP_OK;
}

int tplhip_ilqr_timing(tplhip_ilqr* h, int reset, double* kernel_ms, int64_t* launches)
{
    if (!h) return fail(TPLHIP_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    for (auto& ev : h->events) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second), "event query failed");
        h->kernel_ms += (double)ms;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    h->events.clear();
    if (kernel_ms) *kernel_ms = h->kernel_ms;
    if (launches) *launches = h->launches;
    if (reset) { h->kernel_ms = 0.0; h->launches = 0; }
    h->timing = true;
    return TPLHIP_OK;
}

} // extern "C"
