// Does spreading the independent table look-ups and sin/cos evaluations of ONE ctDynamics call over
// the four lanes of a quad pay on a single wavefront?  Hand-written quad version of the MPC model's
// ctDynamics against the generated one.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../tpl_amd/csrc/ilqr/builtins.h"
#include "../../tpl_amd/models/trajectory_tracking_mpc.inc"

__device__ __forceinline__ double quad_bcast(double v, int q)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (q) {
    case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, true); break;
    case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, true); break;
    case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0xaa, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xaa, 0xf, 0xf, true); break;
    default: lo = __builtin_amdgcn_mov_dpp(lo, 0xff, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xff, 0xf, 0xf, true); break;
    }
    return __hiloint2double(hi, lo);
}

__device__ inline void ctDynamicsQ(const Params& P, const double* x, const double* u, double* out, int q)
{
    // batch 1: four look-ups at x[5], one per lane of the quad
    const DynArr& arr = (q == 0) ? P.ref_phi : (q == 1) ? P.ref_x : (q == 2) ? P.ref_y : P.ref_k;
    const TplInterp it = tpl_init_interp(0.0, P.ref_step, x[5], arr.n0);
    const double p0 = arr.data[it.start], p1 = arr.data[it.end];
    const double lin = it.ia * p0 + it.a * p1;
    const double ang = p0 + tpl_short_angle_dist(p0, p1) * it.a;
    const double mine = (q == 0) ? ang : lin;
    const double w0 = quad_bcast(mine, 0), rx = quad_bcast(mine, 1), ry = quad_bcast(mine, 2), rk = quad_bcast(mine, 3);
    // batch 2: sin/cos of x[2], w0, x[3]
    const double targ = (q == 0) ? x[2] : (q == 1) ? w0 : x[3];
    double ts, tc;
    tpl_sincos(targ, &ts, &tc);
    const double sc1 = quad_bcast(ts, 0), sc2 = quad_bcast(tc, 0);
    const double sc4 = quad_bcast(ts, 1), sc5 = quad_bcast(tc, 1);
    const double s3 = quad_bcast(ts, 2), c3 = quad_bcast(tc, 2);
    out[0] = sc2 * x[4];
    out[1] = sc1 * x[4];
    out[2] = x[4] * (s3 / c3) / (P.l * (1 + (x[4] * x[4]) / (P.v_ch * P.v_ch)));
    out[3] = u[1];
    out[4] = P.a_offset + x[6];
    out[5] = x[4] * (sc1 * sc4 + sc2 * sc5) / (-(-sc4 * (x[0] - rx) + sc5 * (x[1] - ry)) * rk + 1);
    out[6] = u[0];
}

template <int QUAD>
__global__ void k_ct(const Params* Pp, double* xio, long long* cyc, int n_it)
{
    extern __shared__ double sh[];
    Params P = *Pp;
    DynArr* arrs[5] = {&P.ref_x, &P.ref_y, &P.ref_phi, &P.ref_k, &P.ref_v};
    int off = 0;
    for (int a = 0; a < 5; ++a) {
        for (int i = threadIdx.x; i < arrs[a]->n0; i += 64) sh[off + i] = arrs[a]->data[i];
        const double* g = sh + off;
        asm volatile("" : "+v"(g));
        off += (int)arrs[a]->n0;
        arrs[a]->data = g;
    }
    __syncthreads();
    const int cand = QUAD ? threadIdx.x >> 2 : threadIdx.x, q = threadIdx.x & 3;
    double x[7], u[2], k[7];
    for (int i = 0; i < 7; ++i) x[i] = xio[cand * 9 + i];
    u[0] = xio[cand * 9 + 7]; u[1] = xio[cand * 9 + 8];
    long long t0 = clock64();
    for (int it = 0; it < n_it; ++it) {
        if (QUAD) ctDynamicsQ(P, x, u, k, q);
        else ctDynamics(P, x, u, 0.0, 0.05, k, 1);
        for (int i = 0; i < 7; ++i) x[i] = x[i] + k[i] * 1e-4;
    }
    long long t1 = clock64();
    if (!QUAD || q == 0) for (int i = 0; i < 7; ++i) xio[cand * 9 + i] = x[i];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    const int n = 100;
    std::vector<double> rx(n), ry(n), rphi(n), rk(n), rv(n);
    for (int i = 0; i < n; ++i) { rx[i] = 0.5 * i; ry[i] = 0.02 * i; rphi[i] = 0.04; rk[i] = 0.001 * i; rv[i] = 8.0; }
    double* d[5]; std::vector<double>* src[5] = {&rx, &ry, &rphi, &rk, &rv};
    for (int a = 0; a < 5; ++a) { hipMalloc(&d[a], 8 * n); hipMemcpy(d[a], src[a]->data(), 8 * n, hipMemcpyHostToDevice); }
    Params P{};
    P.l = 3.165; P.v_ch = 32.0; P.ref_step = 0.5; P.a_offset = 0.0;
    DynArr* arrs[5] = {&P.ref_x, &P.ref_y, &P.ref_phi, &P.ref_k, &P.ref_v};
    for (int a = 0; a < 5; ++a) { arrs[a]->data = d[a]; arrs[a]->n0 = n; arrs[a]->n1 = 0; arrs[a]->ndim = 1; }
    Params* dP; hipMalloc(&dP, sizeof(Params)); hipMemcpy(dP, &P, sizeof(Params), hipMemcpyHostToDevice);
    std::vector<double> x0(64 * 9);
    for (int l = 0; l < 64; ++l) { double* x = &x0[l * 9]; x[0] = 1.0; x[1] = 0.01 * l; x[2] = 0.001 * l; x[3] = 0.01; x[4] = 8.0; x[5] = 1.0; x[6] = 0.0; x[7] = 0.1; x[8] = 0.01; }
    double* dx; hipMalloc(&dx, 8 * 64 * 9); long long* dc; hipMalloc(&dc, 8);
    std::vector<double> res[2];
    for (int quad = 0; quad < 2; ++quad) for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(dx, x0.data(), 8 * 64 * 9, hipMemcpyHostToDevice);
        const int n_it = 2000;
        if (quad) hipLaunchKernelGGL(k_ct<1>, dim3(1), dim3(32), 8 * 5 * n, 0, dP, dx, dc, n_it);
        else hipLaunchKernelGGL(k_ct<0>, dim3(1), dim3(8), 8 * 5 * n, 0, dP, dx, dc, n_it);
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        res[quad].resize(64 * 9); hipMemcpy(res[quad].data(), dx, 8 * 64 * 9, hipMemcpyDeviceToHost);
        printf("%s: %.0f clock64 ticks per ctDynamics\n", quad ? "quad (8 candidates x 4 lanes)" : "plain (8 lanes)", (double)c / n_it);
    }
    double md = 0;
    for (int c = 0; c < 8; ++c) for (int i = 0; i < 7; ++i) md = fmax(md, fabs(res[0][c * 9 + i] - res[1][c * 9 + i]));
    printf("max |plain - quad| over 8 candidates x 7 states after 2000 steps: %.3e\n", md);
    return 0;
}
