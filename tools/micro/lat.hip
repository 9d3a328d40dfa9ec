// Single-wave latency probes for the iLQR rollout chain (gfx950).  One wave per SIMD is how the
// batched solver runs at batch 1024, so dependent-issue latency is what its chains pay.
//   hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define N_IT 2000

__global__ void k_fma_dep(double* out, long long* cyc, double a, double b)
{
    double x = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x = fma(x, a, b);
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_fma_dep2(double* out, long long* cyc, double a, double b)
{
    double x = out[threadIdx.x], y = out[threadIdx.x + 64];
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x = fma(x, a, b); y = fma(y, b, a); }
    }
    long long t1 = clock64();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_fma32_dep(float* out, long long* cyc, float a, float b)
{
    float x = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x = fmaf(x, a, b);
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_div_dep(double* out, long long* cyc, double a)
{
    double x = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x = a / x + 1.0;
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_lds_chase(int* out, long long* cyc)
{
    __shared__ int tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 7 + 13) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p = tab[p];
    }
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_flat_lds_chase(int* out, long long* cyc)
{
    __shared__ int tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 7 + 13) & 1023;
    __syncthreads();
    // generic pointer to LDS, opaque to the compiler -> flat_load
    const int* g = tab;
    asm volatile("" : "+v"(g));
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p = g[p];
    }
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_global_chase(int* out, long long* cyc, const int* tab)
{
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p = tab[p];
    }
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void k_shfl_dep(double* out, long long* cyc)
{
    double x = out[threadIdx.x];
    long long t0 = clock64();
    for (int i = 0; i < N_IT; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x = __shfl_xor(x, 1 << (j % 6), 64) + 1.0;
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main()
{
    double* d; long long* c; int* tab;
    CK(hipMalloc(&d, 8 * 256)); CK(hipMalloc(&c, 8)); CK(hipMalloc(&tab, 4 * 1024));
    CK(hipMemset(d, 0, 8 * 256));
    int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i * 7 + 13) & 1023;
    CK(hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice));
    long long cy; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* name, int ops_per_it) {
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
        const double n = (double)N_IT * ops_per_it;
        printf("%-34s %8.1f clock64 ticks/op   %8.2f ns/op (kernel wall incl. launch)\n", name, cy / n, ms * 1e6 / n);
    };
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma_dep, dim3(1), dim3(64), 0, 0, d, c, 0.999, 0.001); report("dependent v_fma_f64", 8);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma_dep2, dim3(1), dim3(64), 0, 0, d, c, 0.999, 0.001); report("2 independent chains (per pair)", 8);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma32_dep, dim3(1), dim3(64), 0, 0, (float*)d, c, 0.999f, 0.001f); report("dependent v_fma_f32", 8);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_div_dep, dim3(1), dim3(64), 0, 0, d, c, 1.5); report("dependent f64 divide + add", 4);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, 0, (int*)d, c); report("ds_read_b32 pointer chase", 8);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_flat_lds_chase, dim3(1), dim3(64), 0, 0, (int*)d, c); report("flat_load (LDS aperture) chase", 8);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_global_chase, dim3(1), dim3(64), 0, 0, (int*)d, c, tab); report("global_load chase (L1/L2 hit)", 8);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_shfl_dep, dim3(1), dim3(64), 0, 0, d, c); report("dependent shfl_xor f64 + add", 8);
        printf("--\n");
    }
    return 0;
}
