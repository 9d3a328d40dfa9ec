// Cycles per fp64 sincos / tan / divide / interpolation look-up on ONE wavefront.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include "../../tpl_amd/csrc/ilqr/builtins.h"

template <int WHAT>
__global__ void k(double* io, long long* cyc, int n_it, DynArr arr)
{
    double x = io[threadIdx.x];
    long long t0 = clock64();
    for (int it = 0; it < n_it; ++it) {
        double s, c;
        if (WHAT == 0) { sincos(x, &s, &c); x = x + 1e-3 * (s + c); }
        if (WHAT == 1) { tpl_sincos(x, &s, &c); x = x + 1e-3 * (s + c); }
        if (WHAT == 2) { x = x + 1e-3 * tan(x); }
        if (WHAT == 3) { x = x + 1e-3 * tpl_tan(x); }
        if (WHAT == 4) { x = x + 1e-3 * tpl_lerp(0.0, 0.5, x, arr); }
        if (WHAT == 5) { x = x + 1e-3 * tpl_lerp_angle(0.0, 0.5, x, arr); }
        if (WHAT == 6) { tpl_sincos_core(x, &s, &c); x = x + 1e-3 * (s + c); }
    }
    long long t1 = clock64();
    io[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double* d; long long* dc; double* tab;
    hipMalloc(&d, 8 * 64); hipMalloc(&dc, 8); hipMalloc(&tab, 8 * 100);
    double h[100]; for (int i = 0; i < 100; ++i) h[i] = 0.01 * i;
    hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
    DynArr arr{tab, 100, 0, 1, 0};
    const char* names[7] = {"library sincos", "tpl_sincos", "library tan", "tpl_tan", "tpl_lerp (global array)", "tpl_lerp_angle (global array)", "tpl_sincos_core (no range test)"};
    double x0[64]; for (int l = 0; l < 64; ++l) x0[l] = 0.3 + 0.01 * l;
    const int n_it = 4000;
    for (int w = 0; w < 7; ++w) {
        hipMemcpy(d, x0, sizeof(x0), hipMemcpyHostToDevice);
        switch (w) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, dc, n_it, arr); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, dc, n_it, arr); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, dc, n_it, arr); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, dc, n_it, arr); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, dc, n_it, arr); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d, dc, n_it, arr); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, d, dc, n_it, arr); break;
        }
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("%-34s %7.0f clock64 ticks each (incl. 2 dependent adds/muls)\n", names[w], (double)c / n_it);
    }
    return 0;
}
