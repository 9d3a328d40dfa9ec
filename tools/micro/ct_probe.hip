// Cycles per ctDynamics evaluation of the MPC model on ONE wavefront (how the batched solver
// runs its line-search rollouts at batch 1024).  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-DPROBE_LIBM_TRIG] -o ct_probe ct_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../tpl_amd/csrc/ilqr/builtins.h"
#ifdef PROBE_LIBM_TRIG
#define tpl_sin sin
#define tpl_cos cos
#define tpl_tan tan
#define tpl_sincos(a, s, c) sincos(a, s, c)
#endif
#include "../../tpl_amd/models/trajectory_tracking_mpc.inc"

__global__ void k_ct(const Params* Pp, double* xio, long long* cyc, int n_it, int lds)
{
    extern __shared__ double sh[];
    Params P = *Pp;
    if (lds) {
        // mirror the five reference arrays in LDS and read them through the flat aperture
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
    }
    double x[7], u[2], k[7];
    for (int i = 0; i < 7; ++i) x[i] = xio[threadIdx.x * 9 + i];
    u[0] = xio[threadIdx.x * 9 + 7]; u[1] = xio[threadIdx.x * 9 + 8];
    long long t0 = clock64();
    for (int it = 0; it < n_it; ++it) {
        ctDynamics(P, x, u, 0.0, 0.05, k, 1);
        for (int i = 0; i < 7; ++i) x[i] = x[i] + k[i] * 1e-4;
    }
    long long t1 = clock64();
    for (int i = 0; i < 7; ++i) xio[threadIdx.x * 9 + i] = x[i];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    const int n = 100;
    std::vector<double> rx(n), ry(n), rphi(n), rk(n), rv(n);
    for (int i = 0; i < n; ++i) { rx[i] = 0.5 * i; ry[i] = 0.0; rphi[i] = 0.0; rk[i] = 0.0; rv[i] = 8.0; }
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
    for (int lds = 0; lds < 2; ++lds) for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(dx, x0.data(), 8 * 64 * 9, hipMemcpyHostToDevice);
        const int n_it = 2000;
        hipLaunchKernelGGL(k_ct, dim3(1), dim3(64), 8 * 5 * n, 0, dP, dx, dc, n_it, lds);
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("arrays in %s: %.0f clock64 ticks per ctDynamics\n", lds ? "LDS (flat)" : "global", (double)c / n_it);
    }
    return 0;
}
