// Dependent-load latency of ONE wavefront on gfx950: the same LDS-resident table read (a) through an LDS pointer (ds_read_b64),
// (b) through a generic pointer that points into the LDS aperture (flat_load_dwordx2 -- how the generated iLQR routines read
// their array parameters when the host has placed them in LDS), (c) from global memory (L2 hit).  Each load's address depends on
// the value of the load before (a pointer chase over 512 doubles), 2000 x 8 loads.
//   hipcc --offload-arch=gfx950 -O3 -o lds_flat_lat lds_flat_lat.hip && ./lds_flat_lat
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_IT 2000
typedef __attribute__((address_space(3))) double* LdsD;

template <int MODE>
__global__ void k_chase(const double* __restrict__ tab_g, long long* cyc, double* out)
{
    __shared__ double tab[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = tab_g[i];
    __syncthreads();
    const double* flat = tab;                                    // generic pointer into the LDS aperture
    asm volatile("" : "+v"(flat));                               // (the compiler must not see through it)
    double x = (double)(threadIdx.x & 7);
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = (int)x & 511;
            if (MODE == 0) x = ((LdsD)(unsigned)(unsigned long long)tab)[idx];
            else if (MODE == 1) x = flat[idx];
            else x = tab_g[idx];
        }
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    double h[512];
    for (int i = 0; i < 512; ++i) h[i] = (double)((i * 37 + 11) & 511);
    double *d_tab, *d_out; long long* d_cyc;
    hipMalloc(&d_tab, sizeof h); hipMalloc(&d_out, 64 * sizeof(double)); hipMalloc(&d_cyc, 8);
    hipMemcpy(d_tab, h, sizeof h, hipMemcpyHostToDevice);
    const char* names[3] = {"LDS pointer (ds_read_b64)", "generic pointer into LDS (flat_load)", "global memory, L2 hit (global_load)"};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k_chase<0>, dim3(1), dim3(64), 0, 0, d_tab, d_cyc, d_out);
            if (m == 1) hipLaunchKernelGGL(k_chase<1>, dim3(1), dim3(64), 0, 0, d_tab, d_cyc, d_out);
            if (m == 2) hipLaunchKernelGGL(k_chase<2>, dim3(1), dim3(64), 0, 0, d_tab, d_cyc, d_out);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %7.1f clock64 ticks per dependent load (incl. cvt + and)\n", names[m], (double)c / (N_IT * 8.0));
    }
    return 0;
}
