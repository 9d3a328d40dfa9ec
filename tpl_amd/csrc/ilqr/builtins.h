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

// sin and cos of one argument with a shared range reduction (generated code asks for the
// pair where a routine needs both)
TPL_HD inline void tpl_sincos(double a, double* s, double* c)
{
    sincos(a, s, c);
}

struct DynArr {
    const double* data;   // device pointer in kernels, host pointer on the host
    int64_t n0, n1;       // dims[0], dims[1] (n1 = 0 for 1-D)
    int32_t ndim;
    int32_t pad;
};

// (size_t)floor(q) clamped to [0, n-1] the way optim.c:351 does it
TPL_HD inline int64_t tpl_idx_floor(double q, int64_t n)
{
    const double f = floor(q);
    if (f <= -1.0) return n - 1;                 // negative -> wraps -> min(n-1, huge)
    if (f >= (double)(n - 1)) return n - 1;
    return (int64_t)f;                           // f in (-1, n-1): -0.0 -> 0
}

// (size_t)ceil(q) clamped to [0, n-1] the way optim.c:352 does it
TPL_HD inline int64_t tpl_idx_ceil(double q, int64_t n)
{
    const double c = ceil(q);
    if (c <= -1.0) return n - 1;
    if (c >= (double)(n - 1)) return n - 1;
    return (int64_t)c;                           // c in (-1, 0] -> 0
}

struct TplInterp { double a, ia; int64_t start, end; };

// initInterp, optim.c:347-355
TPL_HD inline TplInterp tpl_init_interp(double x0, double dx, double x, int64_t n)
{
    TplInterp i;
    const double q = (x - x0) / dx;
    i.start = tpl_idx_floor(q, n);
    i.end = tpl_idx_ceil(q, n);
    double a = q - (double)i.start;
    a = a > 0.0 ? a : 0.0;                       // min(max(0.0, .), 1.0)
    a = a < 1.0 ? a : 1.0;
    i.a = a;
    i.ia = 1.0 - a;
    return i;
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
    return (fabs(x) < m) ? x : fmod(x, m);
}

// shortAngleDist, optim.c:332-338
TPL_HD inline double tpl_short_angle_dist(double a0, double a1)
{
    const double m = M_PI * 2;
    const double da = tpl_fmod_pos(a1 - a0, m);
    return tpl_fmod_pos(2 * da, m) - da;
}

// __lerp_angle, optim.c:394-411
TPL_HD inline double tpl_lerp_angle(double x0, double dx, double x, const DynArr& arr)
{
    const int64_t n = arr.n0;
    if (n == 0) return 0.0;
    const TplInterp i = tpl_init_interp(x0, dx, x, n);
    return arr.data[i.start] + tpl_short_angle_dist(arr.data[i.start], arr.data[i.end]) * i.a;
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
