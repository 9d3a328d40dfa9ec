// Vector-memory issue-rate probe (gfx950): cycles per wave-level buffer_load of 4 / 8 / 12 / 16 bytes per lane when the
// data sits in the CU's L1 (a 4 KB window per wave re-read) or in L2 (a per-wave stride walk over 4 MB), at 8 waves
// per SIMD.  Prices the row loads of the DP sweep: is the texture-address path bound by lanes, by instructions or by bytes?
//   hipcc --offload-arch=gfx950 -O3 -o vmem_tp vmem_tp.hip && ./vmem_tp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITERS 2048

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));

// STRIDE: bytes between consecutive lanes (0 = BYTES, a packed array; 16 with BYTES = 12: three of every four dwords of
// a 16-byte cell array -- does a quad of lanes that never straddles a 64-byte sector cost the L1 fewer accesses?)
template <int BYTES, bool L2WALK, int STRIDE = 0>
__global__ void __launch_bounds__(256) k_load(const float* __restrict__ src, float* __restrict__ out, unsigned span_bytes)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)span_bytes, 0x00020000);
    constexpr int LS = STRIDE ? STRIDE : BYTES;
    const unsigned lane_off = (threadIdx.x & 63) * LS;
    // L1: every wave re-reads its own 8 x (64 x BYTES) window.  L2: the window start walks through span_bytes.
    unsigned base = L2WALK ? ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 * 64 * LS) % (span_bytes - 16 * 64 * LS) : 0u;   // L1: one 8-load window for all
    float acc = 0.0f;
    unsigned lo = lane_off;
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("" : "+v"(lo));      // keeps the (L1-resident) loads inside the loop
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned so = base + j * 64 * LS;
            if constexpr (BYTES == 4)  { acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lo, so, 0)); }
            if constexpr (BYTES == 8)  { v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, lo, so, 0)); acc += v.x; }
            if constexpr (BYTES == 12) { v3f v = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, lo, so, 0)); acc += v.x; }
            if constexpr (BYTES == 16) { v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, lo, so, 0)); acc += v.x; }
        }
        if (L2WALK) base = (base + 1031u * 64 * LS) % (span_bytes - 16 * 64 * LS);
    }
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, wps = 8, blocks = cus * wps;
    const unsigned span = 24u << 20;
    float *src, *out; CK(hipMalloc(&src, span)); CK(hipMalloc(&out, 4096)); CK(hipMemset(src, 0, span));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, unsigned sp, int bytes) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, out, sp);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double inst_per_cu = (double)ITERS * 8 * wps * 4;     // wave-instructions per CU
        const double ns = ms * 1e6 / inst_per_cu;
        printf("%-22s %8.3f ms  %6.2f ns per wave-load per CU = %5.1f cycles at 2.4 GHz; %6.1f B/clk/CU; %5.2f TB/s chip\n", name, ms, ns,
               ns * 2.4, 64.0 * bytes / (ns * 2.4), 64.0 * bytes * inst_per_cu * cus / (ms * 1e-3) / 1e12);
    };
    run("L1 dword", k_load<4, false>, span, 4);   run("L1 dwordx2", k_load<8, false>, span, 8);
    run("L1 dwordx3", k_load<12, false>, span, 12); run("L1 dwordx4", k_load<16, false>, span, 16);
    const unsigned l2 = 2u << 20;
    run("L2 dword", k_load<4, true>, l2, 4);   run("L2 dwordx2", k_load<8, true>, l2, 8);
    run("L2 dwordx3", k_load<12, true>, l2, 12); run("L2 dwordx4", k_load<16, true>, l2, 16);
    run("MALL dwordx3", k_load<12, true>, span, 12); run("MALL dwordx4", k_load<16, true>, span, 16);
    run("L1 dwordx3 stride16", k_load<12, false, 16>, span, 12); run("L2 dwordx3 stride16", k_load<12, true, 16>, l2, 12);
    run("L1 dwordx2 stride16", k_load<8, false, 16>, span, 8); run("L1 dword stride16", k_load<4, false, 16>, span, 4);
    run("L1 dwordx3 stride24", k_load<12, false, 24>, span, 12); run("L1 dwordx3 stride32", k_load<12, false, 32>, span, 12);
    return 0;
}
