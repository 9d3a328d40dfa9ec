// What does SQ_LDS_BANK_CONFLICT count on gfx950?  One wavefront per workgroup (the solver's shape), N_IT x 8 LDS reads per lane
// in a handful of address patterns, one kernel per pattern so that rocprofv3 --pmc lists them apart:
//   b32_seq      ds_read_b32,  lane i reads dword i                         (the conflict-free reference)
//   b64_seq      ds_read_b64,  lane i reads double i                        (64 x 8 B = 512 B per instruction)
//   b64_bcast    ds_read_b64,  every lane reads the same double
//   b64_col9     ds_read_b64,  lane i reads double (i % 9): nine distinct addresses, broadcast among lanes -- phase A / B of the
//                              Riccati stage (a column of [fx fu])
//   b64_row7     ds_read_b64,  lane i reads double 7 * (i / 9): eight distinct addresses 7 doubles apart -- Vxx rows in phase A
//   b64_stride2  ds_read_b64,  lane i reads double 2 i                      (a true two-way conflict for comparison)
//   b64_stride16 ds_read_b64,  lane i reads double 16 i                     (16 doubles = 128 B apart: every lane the same bank pair)
//   b128_seq     ds_read_b128, lane i reads 16 bytes at 16 i
//   ring56       ds_write_b64, lane i writes double 56 * (i / 4) + (i % 4)  (rollout ring: 8 candidates x quad, stage stride 56)
//   hipcc --offload-arch=gfx950 -O3 -o lds_conflict_probe lds_conflict_probe.hip
//   rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d out -- ./lds_conflict_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_IT 4000
typedef __attribute__((address_space(3))) double* LdsD;
typedef __attribute__((address_space(3))) float* LdsF;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d* LdsD2;

__device__ __forceinline__ int pattern(int mode, int lane)
{
    switch (mode) {
    case 0: return lane;                 // dword index
    case 1: return lane;                 // double index from here on
    case 2: return 0;
    case 3: return lane % 9;
    case 4: return 7 * (lane / 9);
    case 5: return 2 * lane;
    case 6: return 16 * lane;
    case 7: return 2 * lane;             // b128: double index of the pair
    default: return 56 * (lane / 4) + (lane % 4);
    }
}

template <int MODE>
__global__ void __launch_bounds__(64) k_probe(double* out)
{
    __shared__ double tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) tab[i] = (double)i;
    __syncthreads();
    const int idx = pattern(MODE, (int)threadIdx.x);
    double acc = 0.0;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (the offset j * 64 doubles keeps the pattern and stops the compiler from merging the reads)
            if (MODE == 0) acc += (double)((LdsF)(unsigned)(unsigned long long)tab)[idx + j * 128];
            else if (MODE == 7) { const v2d v = ((LdsD2)(unsigned)(unsigned long long)tab)[idx / 2 + j * 32]; acc += v.x + v.y; }
            else if (MODE == 8) { ((LdsD)(unsigned)(unsigned long long)tab)[idx + j * 4 * 0 + ((it + j) & 7) * 7] = acc; acc += 1.0; }
            else acc += ((LdsD)(unsigned)(unsigned long long)tab)[idx + j * 64];
        }
        asm volatile("" : "+v"(acc));
    }
    out[threadIdx.x + 64 * blockIdx.x] = acc + tab[threadIdx.x];
}

int main()
{
    double* d_out;
    hipMalloc(&d_out, 64 * 1024 * sizeof(double));
    const dim3 grid(1024), block(64);
    hipLaunchKernelGGL(k_probe<0>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<1>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<2>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<3>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<4>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<5>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<6>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<7>, grid, block, 0, 0, d_out);
    hipLaunchKernelGGL(k_probe<8>, grid, block, 0, 0, d_out);
    hipDeviceSynchronize();
    printf("modes: 0 b32_seq, 1 b64_seq, 2 b64_bcast, 3 b64_col9, 4 b64_row7, 5 b64_stride2, 6 b64_stride16, 7 b128_seq, 8 ring56 (writes)\n");
    return 0;
}
