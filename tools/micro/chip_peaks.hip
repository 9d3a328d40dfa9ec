// The chip figures the roofline fractions are quoted against, measured on the box ("to be re-measured on the
// box with a copy/triad and an FMA microbenchmark before fractions are quoted"): HBM read / copy / triad bandwidth on
// buffers several times the 256 MiB Infinity Cache, and the vector FMA rate in fp32, packed fp32 and fp64 with eight
// wavefronts per SIMD and eight independent accumulators per lane.
//   hipcc --offload-arch=gfx950 -O3 -o chip_peaks tools/micro/chip_peaks.hip && ./chip_peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_read(const v4f* __restrict__ a, size_t n, float* out)
{
    v4f acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

__global__ void __launch_bounds__(256) k_copy(const v4f* __restrict__ a, v4f* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

__global__ void __launch_bounds__(256) k_triad(v4f* __restrict__ a, const v4f* __restrict__ b, const v4f* __restrict__ c, float s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = b[i] + s * c[i];
}

template <typename T>
__global__ void __launch_bounds__(256) k_fma(T* out, T x, T y, int iters)
{
    T a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = a0 * y + x; a1 = a1 * y + x; a2 = a2 * y + x; a3 = a3 * y + x;
            a4 = a4 * y + x; a5 = a5 * y + x; a6 = a6 * y + x; a7 = a7 * y + x;
        }
    }
    const T s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == (T)123.456) out[threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_pk_fma(float* out, float x, float y, int iters)
{
    v2f a0 = {x, x + 1}, a1 = {x + 2, x + 3}, a2 = {x + 4, x + 5}, a3 = {x + 6, x + 7};
    v2f a4 = {x + 8, x + 9}, a5 = {x + 10, x + 11}, a6 = {x + 12, x + 13}, a7 = {x + 14, x + 15};
    const v2f yy = {y, y}, xx = {x, x};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_elementwise_fma(a0, yy, xx); a1 = __builtin_elementwise_fma(a1, yy, xx);
            a2 = __builtin_elementwise_fma(a2, yy, xx); a3 = __builtin_elementwise_fma(a3, yy, xx);
            a4 = __builtin_elementwise_fma(a4, yy, xx); a5 = __builtin_elementwise_fma(a5, yy, xx);
            a6 = __builtin_elementwise_fma(a6, yy, xx); a7 = __builtin_elementwise_fma(a7, yy, xx);
        }
    }
    const v2f s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s.x + s.y == 123.456f) out[threadIdx.x] = s.x;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock rate prop %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    const size_t bytes = (size_t)2 << 30;                   // per array: 2 GiB = 8 x the Infinity Cache
    const size_t n = bytes / sizeof(v4f);
    v4f *a, *b, *c;
    float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    for (int wg_per_cu : {8, 16, 32}) {
        const int grid = cus * wg_per_cu;
        const double r = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); }, 5);
        const double cp = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 5);
        const double tr = time_ms([&] { hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, 0, a, b, c, 1.5f, n); }, 5);
        printf("HBM, %2d workgroups per CU: read %6.2f TB/s   copy %6.2f TB/s (read + write)   triad %6.2f TB/s (2 reads + 1 write)\n", wg_per_cu,
               bytes / r / 1e9, 2.0 * bytes / cp / 1e9, 3.0 * bytes / tr / 1e9);
    }
    const int iters = 2048, grid = cus * 8;                  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    const double lanes = (double)grid * 256;
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_fma<float>, dim3(grid), dim3(256), 0, 0, out, 1.0f, 0.999f, iters); }, 5);
        printf("fp32 v_fma_f32     : %7.1f TFLOP/s\n", lanes * iters * 64.0 * 2.0 / ms / 1e9);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_pk_fma, dim3(grid), dim3(256), 0, 0, out, 1.0f, 0.999f, iters); }, 5);
        printf("fp32 v_pk_fma_f32  : %7.1f TFLOP/s\n", lanes * iters * 64.0 * 4.0 / ms / 1e9);
    }
    {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_fma<double>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<double*>(out), 1.0, 0.999, iters); }, 5);
        printf("fp64 v_fma_f64     : %7.1f TFLOP/s\n", lanes * iters * 64.0 * 2.0 / ms / 1e9);
    }
    return 0;
}
