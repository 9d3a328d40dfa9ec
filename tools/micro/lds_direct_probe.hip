// lds_direct_probe.hip -- where does gfx950's `buffer_load_dwordx3 ... lds` put the 12 bytes of lane i?
// (The ISA text this image ships does not say for the three- and four-dword forms.)  Every lane of a 256-thread workgroup
// loads its 12 bytes of a float ramp straight into LDS; the kernel then dumps the LDS block.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/lds_direct_probe tools/micro/lds_direct_probe.hip && tools/micro/bin/lds_direct_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* __restrict__ src, float* __restrict__ out, int n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* f = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) f[i] = -1.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n * 4, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // one LDS base per wavefront (it travels in M0), one 12-byte element per lane
    __attribute__((address_space(3))) void* lds = (__attribute__((address_space(3))) void*)(smem + wave * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 12, (wave * 64 + lane) * 12, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = f[i];
}

int main()
{
    const int n = 4096;
    std::vector<float> h(n), o(1024);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *dout;
    hipMalloc(&d, n * 4);
    hipMalloc(&dout, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, dout, n);
    hipMemcpy(o.data(), dout, 1024 * 4, hipMemcpyDeviceToHost);
    // expectation A: lane i of wave w writes floats [3(64w+i), +3) at LDS float index 256 w + 3 i
    int ok = 1;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int c = 0; c < 3; ++c)
                if (o[256 * w + 3 * l + c] != (float)(3 * (64 * w + l) + c)) ok = 0;
    std::printf("lane-contiguous 12-byte elements behind the wave's M0 base: %s\n", ok ? "yes" : "NO");
    if (!ok) for (int i = 0; i < 40; ++i) std::printf("lds[%d] = %g\n", i, o[i]);
    return ok ? 0 : 1;
}
