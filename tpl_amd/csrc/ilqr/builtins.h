// builtins.h -- run-time semantics of the opaque functions of tpl_amd.optim.symext,
// callable from host and device.  They restate the interpolation helpers of the
// reference's solver template (library/tpl/optim/templates/optim.c:297-486),
// including their edge behaviour, because generated derivative code evaluates
// them at x +- 5e-5 / +- 1e-4 right at the table borders:
//
//  * index clamp: the template casts floor(q) / ceil(q) to size_t before
//    clamping (initInterp :347-355).  A negative value wraps to a huge unsigned
//    number and is then clamped to the LAST index, so for q < 0 the lookup
//    returns arr[n-1] (start = n-1, weight 0).  Reproduced explicitly here
//    (idx_floor / idx_ceil) instead of through an out-of-range cast.
//  * an empty array reads as 0 (:369-371).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TPL_HD __host__ __device__
#else
#define TPL_HD
#endif

// ---- sin / cos / tan ------------------------------------------------------
// The line-search rollouts are chains of dependent dynamics evaluations on a
// single wavefront, so what the trigonometric calls cost is their INSTRUCTION
// COUNT: the device library's fp64 sincos is ~95 instructions on its small-argument
// path and tan ~115, against ~35 here for one shared reduction plus both kernels.
//   k = rint(a * 2/pi);  r = a - k*pi/2 in two FMAs (hi part exact for |a| < 1e5, the
//   lo part rounds once);  sin r, cos r by the classic degree-13/14 minimax polynomials
//   on |r| <= pi/4 (coefficients of the public-domain fdlibm __kernel_sin/__kernel_cos);
//   quadrant by k mod 4.
// Accuracy ~1 ulp (tan ~2.5 ulp as sin/cos), far inside the solver's 1e-5 contract; the
// CPU oracle uses the C library's functions and does not share this code.  |a| >= 1e5,
// infinities and NaNs go to the library.
// fma(a, b, c) as the THREE-address instruction.  Left to the compiler, a Horner step whose addend is a constant held in a
// register pair becomes v_mov_b64 + v_fmac_f64 (the two-address form wants the addend in the destination): two instructions for
// one, ten times per sin/cos pair.  On a lone wavefront the moves hide in the fma chain's latency (nothing at 1 024 problems); with two
// wavefronts per SIMD issue slots count: 9.45 -> 9.60 M problem-iterations/s at 4 096, same box (profiles/r05_ilqr_guards_ab.txt).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double tpl_fma3(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#else
TPL_HD inline double tpl_fma3(double a, double b, double c) { return fma(a, b, c); }
#endif

TPL_HD inline void tpl_sincos_core(double a, double* s, double* c)
{
    const double k = rint(a * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, a);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = tpl_fma3(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = tpl_fma3(z, ps, 2.75573137070700676789e-06);
    ps = tpl_fma3(z, ps, -1.98412698298579493134e-04);
    ps = tpl_fma3(z, ps, 8.33333333332248946124e-03);
    ps = tpl_fma3(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    double pc = tpl_fma3(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = tpl_fma3(z, pc, -2.75573143513906633035e-07);
    pc = tpl_fma3(z, pc, 2.48015872894767294178e-05);
    pc = tpl_fma3(z, pc, -1.38888888888741095749e-03);
    pc = tpl_fma3(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + (z * z) * pc);
    const int n = (int)k;
    const double ss = (n & 1) ? cr : sr;
    const double cc = (n & 1) ? sr : cr;
    *s = (n & 2) ? -ss : ss;
    *c = ((n + 1) & 2) ? -cc : cc;
}

// Rare slow paths are guarded by a WAVE-UNIFORM test on the device: a per-lane `if` around a
// library call costs ~135 cycles per call site in exec-mask bookkeeping and result merging even
// when no lane takes it (measured, tools/micro/trig_probe.hip); a ballot is one compare and a
// scalar branch.
#if defined(__HIP_DEVICE_COMPILE__)
#define TPL_ANY_LANE(cond) (__builtin_amdgcn_ballot_w64(cond) != 0ull)
// First statement of the block behind `if (TPL_ANY_LANE(c))`: with it the test is a compare and `s_cbranch_vccz`, two
// instructions where no lane takes the path; without it the compiler folds the uniform test into the per-lane one behind it --
// compare, s_cmp, s_cselect, s_and, s_and_saveexec, branch, s_or: seven.  A rollout stage passes eight such tests, and on a
// lone wavefront an instruction of any class is four cycles (round 5: 6.33 -> 6.50 k batched iterations/s at 1 024 x N = 50,
// profiles/r05_ilqr_guards_ab.txt).
#define TPL_SLOW_PATH() asm volatile("; slow path")
#else
#define TPL_ANY_LANE(cond) (cond)
#define TPL_SLOW_PATH() do { } while (0)
#endif

TPL_HD inline void tpl_sincos(double a, double* s, double* c)
{
    tpl_sincos_core(a, s, c);
    if (TPL_ANY_LANE(!(fabs(a) < 1.0e5))) {
        TPL_SLOW_PATH();
        if (!(fabs(a) < 1.0e5)) sincos(a, s, c);
    }
}

TPL_HD inline double tpl_sin(double a) { double s, c; tpl_sincos(a, &s, &c); return s; }
TPL_HD inline double tpl_cos(double a) { double s, c; tpl_sincos(a, &s, &c); return c; }

TPL_HD inline double tpl_tan(double a)
{
    double s, c;
    tpl_sincos_core(a, &s, &c);
    double r = s / c;
    if (TPL_ANY_LANE(!(fabs(a) < 1.0e5))) {
        TPL_SLOW_PATH();
        if (!(fabs(a) < 1.0e5)) r = tan(a);
    }
    return r;
}

struct DynArr {
    const double* data;   // device pointer in kernels, host pointer on the host
    int64_t n0, n1;       // dims[0], dims[1] (n1 = 0 for 1-D)
    int32_t ndim;
    int32_t pad;
};

// (size_t)floor(q) clamped to [0, n-1] the way optim.c:351 does it: a negative value wraps to a
// huge unsigned number and clamps to n-1.  The clamp is done on the (integral) double so the
// index needs one f64->i32 conversion and the weight no conversion back (there is no single
// f64<->i64 instruction; array lengths are far below 2^31).
TPL_HD inline double tpl_idxf(double f, double nm1)
{
    return (f <= -1.0 || f >= nm1) ? nm1 : f;    // f in (-1, n-1): -0.0 -> index 0
}

TPL_HD inline int64_t tpl_idx_floor(double q, int64_t n)
{
    return (int64_t)(int)tpl_idxf(floor(q), (double)(n - 1));
}

// (size_t)ceil(q) clamped to [0, n-1] the way optim.c:352 does it
TPL_HD inline int64_t tpl_idx_ceil(double q, int64_t n)
{
    return (int64_t)(int)tpl_idxf(ceil(q), (double)(n - 1));
}

struct TplInterp { double a, ia; int start, end; };

// initInterp, optim.c:347-355
TPL_HD inline TplInterp tpl_init_interp(double x0, double dx, double x, int64_t n)
{
    TplInterp i;
    const double q = (x - x0) / dx;
    const double nm1 = (double)(int)(n - 1);
    const double fs = tpl_idxf(floor(q), nm1);
    const double fe = tpl_idxf(ceil(q), nm1);
    i.start = (int)fs;
    i.end = (int)fe;
    double a = q - fs;                           // == q - (double)start
    a = a > 0.0 ? a : 0.0;                       // min(max(0.0, .), 1.0)
    a = a < 1.0 ? a : 1.0;
    i.a = a;
    i.ia = 1.0 - a;
    return i;
}

// a / b for a divisor that does not change inside the loop this is inlined into (the stride of a look-up table across the stages
// of a rollout): r = 1 / b is an IEEE division the compiler moves out of the loop, and q = fma(fma(-a r, b, a), r, a r) is the
// correctly rounded quotient given the correctly rounded reciprocal (Markstein) -- three dependent instructions on the chain
// instead of the twelve of v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup.  Checked against `/` on 4e8 random pairs
// incl. exact and near-exact multiples: equal wherever a is 0 or a normal number (a subnormal dividend can lose its last bit).
// Used by the quad look-up of the rollouts only -- inside a routine that runs once per lane the reciprocal would be one more
// division, not one less.  Same box: 6.85 -> 6.96 k batched iterations/s at 1 024 x N = 50 (profiles/r05_ilqr_guards_ab.txt).
TPL_HD inline double tpl_div_invariant(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r = 1.0 / b;
    const double q0 = a * r;
    double q = fma(fma(-q0, b, a), r, q0);
    // Outside the domain of that identity the correction turns inf into NaN (a = +-inf: a line-search candidate that diverged;
    // b = 0) -- the look-up would then read data[0] where `/` reads data[n - 1], which is what the oracle and the reference do.
    // One compare and a branch no lane takes on the fast path (the guard idiom of this file); a NaN dividend takes it too.
    // (A SUBNORMAL dividend can lose its last bit here: its quotient is an interpolation weight below 2^-1022 either way.)
    if (TPL_ANY_LANE(!(fabs(q0) < 1.0e300))) {
        TPL_SLOW_PATH();
        if (!(fabs(q0) < 1.0e300)) q = a / b;
    }
    return q;
#else
    return a / b;
#endif
}

// get_array_value, optim.c:330
TPL_HD inline double tpl_get_array_value(const DynArr& arr, double i)
{
    return arr.data[(int64_t)i];
}

// __box_interp, optim.c:357-373
TPL_HD inline double tpl_box_interp(double dx, double x, const DynArr& arr)
{
    const int64_t n = arr.n0;
    if (n == 0) return 0.0;
    return arr.data[tpl_idx_floor(x / dx, n)];
}

// __lerp, optim.c:375-392
TPL_HD inline double tpl_lerp(double x0, double dx, double x, const DynArr& arr)
{
    const int64_t n = arr.n0;
    if (n == 0) return 0.0;
    const TplInterp i = tpl_init_interp(x0, dx, x, n);
    return i.ia * arr.data[i.start] + i.a * arr.data[i.end];
}

// fmod(x, m) for m > 0: |x| < m returns x itself (that IS fmod's exact result); only larger
// arguments take the library's long-division loop, which is very slow in fp64 on the GPU.
TPL_HD inline double tpl_fmod_pos(double x, double m)
{
    double r = x;
    if (TPL_ANY_LANE(!(fabs(x) < m))) {
        TPL_SLOW_PATH();
        if (!(fabs(x) < m)) r = fmod(x, m);
    }
    return r;
}

// shortAngleDist, optim.c:332-338
TPL_HD inline double tpl_short_angle_dist(double a0, double a1)
{
    const double m = M_PI * 2;
    // ONE test for both remainders: below pi in magnitude the difference is its own remainder and so is its double
    // (|2 d| < 2 pi), i.e. the two fmod calls return their arguments -- the same two operations, the same bits
    const double d = a1 - a0;
    if (TPL_ANY_LANE(!(fabs(d) < M_PI))) {
        TPL_SLOW_PATH();
        const double da = tpl_fmod_pos(d, m);
        return tpl_fmod_pos(2 * da, m) - da;
    }
    return 2 * d - d;
}

// __lerp_angle, optim.c:394-411
TPL_HD inline double tpl_lerp_angle(double x0, double dx, double x, const DynArr& arr)
{
    const int64_t n = arr.n0;
    if (n == 0) return 0.0;
    const TplInterp i = tpl_init_interp(x0, dx, x, n);
    return arr.data[i.start] + tpl_short_angle_dist(arr.data[i.start], arr.data[i.end]) * i.a;
}

// Several __lerp / __lerp_angle look-ups at ONE abscissa (the generator batches them): the
// division, floor and ceil are shared, every table is read before the first value is used, and per
// table the arithmetic is exactly tpl_lerp's / tpl_lerp_angle's (bit i of angle_mask picks the
// latter).  An empty table yields 0.0 like the builtins (p0 = p1 = a = 0 gives exactly that).
#if defined(__HIP_DEVICE_COMPILE__)
#define TPL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TPL_SCHED_FENCE() do { } while (0)
#endif
template <int N>
TPL_HD inline void tpl_lerp_batch(double x0, double dx, double x, const DynArr* const (&arrs)[N], unsigned angle_mask,
                                  double (&out)[N])
{
    const double q = (x - x0) / dx;
    const double fq = floor(q), cq = ceil(q);
    double a[N], p0[N], p1[N];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < N; ++i) {
        const int64_t n = arrs[i]->n0;
        a[i] = 0.0; p0[i] = 0.0; p1[i] = 0.0;
        if (n != 0) {
            const double nm1 = (double)(int)(n - 1);
            const double fs = tpl_idxf(fq, nm1), fe = tpl_idxf(cq, nm1);
            double w = q - fs;
            w = w > 0.0 ? w : 0.0;
            w = w < 1.0 ? w : 1.0;
            a[i] = w;
            p0[i] = arrs[i]->data[(int)fs];
            p1[i] = arrs[i]->data[(int)fe];
        }
    }
    TPL_SCHED_FENCE();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < N; ++i) {
        if ((angle_mask >> i) & 1u) out[i] = p0[i] + tpl_short_angle_dist(p0[i], p1[i]) * a[i];
        else out[i] = (1.0 - a[i]) * p0[i] + a[i] * p1[i];
    }
}

// __lerp_wrap, optim.c:413-456 (no index clamping in the template; callers keep x in range)
TPL_HD inline double tpl_lerp_wrap(double len, double dx, double x, const DynArr& xs, const DynArr& arr)
{
    const int64_t n = arr.n0;
    if (n == 0) return 0.0;
    const double first = xs.data[0];
    const double last = first + (double)(n - 1) * dx;
    const double gap = len - (last - first);
    x = fmod(x - first, len);
    if (x < 0) x += len;
    x += first;
    double alpha;
    int64_t start, end;
    if (x >= last && gap > 0) {
        alpha = (x - last) / gap;
        start = n - 1;
        end = 0;
    } else {
        const double q = (x - first) / dx;
        start = (int64_t)floor(q);
        end = (int64_t)ceil(q);
        if (start > n - 1) start = n - 1;        // guard reads past the end (template: undefined)
        if (end > n - 1) end = n - 1;
        if (start < 0) start = 0;
        if (end < 0) end = 0;
        alpha = q - (double)start;
    }
    return (1.0 - alpha) * arr.data[start] + alpha * arr.data[end];
}

// __blerp, optim.c:458-486: arr[rows = y][cols = x]
TPL_HD inline double tpl_blerp(double x0, double y0, double dx, double dy, double x, double y, const DynArr& arr)
{
    const int64_t rows = arr.n0, cols = arr.n1;
    if (rows == 0 || cols == 0) return 0.0;
    const TplInterp ix = tpl_init_interp(x0, dx, x, cols);
    const TplInterp iy = tpl_init_interp(y0, dy, y, rows);
    const double p0 = iy.ia * arr.data[iy.start * cols + ix.start] + iy.a * arr.data[iy.end * cols + ix.start];
    const double p1 = iy.ia * arr.data[iy.start * cols + ix.end] + iy.a * arr.data[iy.end * cols + ix.end];
    return ix.ia * p0 + ix.a * p1;
}

// ---- quad-cooperative helpers (generated <routine>Q code, device only) -----------------------
// Four neighbouring lanes (a "quad") hold the same arguments; each evaluates one of up to four
// independent sub-expressions and the results are exchanged with DPP quad-permute moves (two
// 32-bit moves per double, no LDS round trip).
#if defined(__HIPCC__)
__device__ inline int tpl_quad_lane() { return (int)(__lane_id() & 3u); }

__device__ inline double tpl_quad_bcast(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (src) {      // the DPP control is an immediate: quad_perm:[src, src, src, src]
    case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true); break;
    case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, true); break;
    case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0xaa, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xaa, 0xf, 0xf, true); break;
    default: lo = __builtin_amdgcn_mov_dpp(lo, 0xff, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xff, 0xf, 0xf, true); break;
    }
    return __hiloint2double(hi, lo);
}

__device__ inline double tpl_quad_pick(int q, double a0, double a1, double a2, double a3)
{
    // selects on the halves (as a ?: chain on doubles the compiler builds a switch: masked branches, eight instructions)
    int lo = __double2loint(a3), hi = __double2hiint(a3);
    lo = (q == 2) ? __double2loint(a2) : lo; hi = (q == 2) ? __double2hiint(a2) : hi;
    lo = (q == 1) ? __double2loint(a1) : lo; hi = (q == 1) ? __double2hiint(a1) : hi;
    lo = (q == 0) ? __double2loint(a0) : lo; hi = (q == 0) ? __double2hiint(a0) : hi;
    return __hiloint2double(hi, lo);
}

// lane q of the quad: tpl_lerp (or tpl_lerp_angle where bit q of angle_mask is set) of arrs[q]
__device__ inline double tpl_quad_lookup(int q, double x0, double dx, double x, const DynArr& a0, const DynArr& a1,
                                         const DynArr& a2, const DynArr& a3, unsigned angle_mask)
{
    // Four separate references and a select over the VALUES (length, base pointer): the descriptors
    // stay plain member reads of the parameter record, which may then live in registers.  (An array
    // of descriptor pointers indexed by the lane ends up in a stack slot.)
    const int64_t n0 = a0.n0, n1 = a1.n0, n2 = a2.n0, n3 = a3.n0;         // read all four first: a lazy ?:
    const double* const d0 = a0.data; const double* const d1 = a1.data;   // chain is turned back into ONE
    const double* const d2 = a2.data; const double* const d3 = a3.data;   // load through a selected address
    const int64_t n = (q == 0) ? n0 : (q == 1) ? n1 : (q == 2) ? n2 : n3;
    const double* data = (q == 0) ? d0 : (q == 1) ? d1 : (q == 2) ? d2 : d3;
    if (n == 0) return 0.0;
    TplInterp i;                                                          // tpl_init_interp with the division by the invariant stride
    {
        const double qq = tpl_div_invariant(x - x0, dx);
        const double nm1 = (double)(int)(n - 1);
        const double fs = tpl_idxf(floor(qq), nm1), fe = tpl_idxf(ceil(qq), nm1);
        i.start = (int)fs; i.end = (int)fe;
        double aa = qq - fs;
        aa = aa > 0.0 ? aa : 0.0;
        aa = aa < 1.0 ? aa : 1.0;
        i.a = aa; i.ia = 1.0 - aa;
    }
    const double p0 = data[i.start], p1 = data[i.end];
    const double lin = i.ia * p0 + i.a * p1;                              // tpl_lerp
    if (angle_mask == 0u) return lin;
    const double ang = p0 + tpl_short_angle_dist(p0, p1) * i.a;           // tpl_lerp_angle
    return ((angle_mask >> q) & 1u) ? ang : lin;
}

// tan(a) from the (sin a, cos a) pair a quad batch produced: what tpl_tan computes
__device__ inline double tpl_tan_sc(double a, double s, double c)
{
    double r = s / c;
    if (TPL_ANY_LANE(!(fabs(a) < 1.0e5))) {
        TPL_SLOW_PATH();
        if (!(fabs(a) < 1.0e5)) r = tan(a);
    }
    return r;
}
#endif

