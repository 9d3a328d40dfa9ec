// FETCH_SIZE / WRITE_SIZE calibration for the DP sweep's access shapes (gfx950).
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for 16 B/lane streaming reads (reports exactly half the bytes);
// the sweep reads its value tables with buffer_load_dwordx3 (12 B/lane, a 768-byte run per wave-instruction, runs of
// one table row of S x 12 bytes) and stores them with 12-byte stores.  Each kernel here moves a KNOWN number of bytes
// ONCE from a buffer several times larger than the 256 MiB Infinity Cache (so every byte comes from / goes to HBM);
// one `rocprofv3 --pmc FETCH_SIZE` pass (and one WRITE_SIZE pass) then gives the factor per shape:
//   hipcc --offload-arch=gfx950 -O3 -o fetch_cal tools/micro/fetch_cal.hip
//   rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d out -- ./fetch_cal
//   python3 tools/fetch_cal_summary.py out     -> bytes moved / (FETCH_SIZE x 1024) per kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));

// every wave streams whole "rows" of ROW_LANES x BYTES (the sweep: 256 lanes x 12 B = one 3 KiB table row, four waves)
template <int BYTES>
__global__ void __launch_bounds__(256) k_read(const char* __restrict__ src, float* __restrict__ out, unsigned rows, unsigned row_bytes)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)0x7fffffff, 0x00020000);
    const unsigned lane_off = threadIdx.x * BYTES;
    float acc = 0.0f;
    for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
        const unsigned so = row * row_bytes;
        if constexpr (BYTES == 4)  acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, so, 0));
        if constexpr (BYTES == 12) { v3f v = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, lane_off, so, 0)); acc += v.x + v.y + v.z; }
        if constexpr (BYTES == 16) { v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, so, 0)); acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int BYTES>
__global__ void __launch_bounds__(256) k_write(char* __restrict__ dst, unsigned rows, unsigned row_bytes)
{
    for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
        char* p = dst + (size_t)row * row_bytes + threadIdx.x * BYTES;
        const float x = (float)row;
        if constexpr (BYTES == 12) {      // the sweep's st_node: a 12-byte store per lane
            *reinterpret_cast<v3f*>(p) = (v3f){x, x, x};
        }
        if constexpr (BYTES == 16) *reinterpret_cast<v4f*>(p) = (v4f){x, x, x, x};
    }
}

int main()
{
    const size_t span = (size_t)1536 << 20;          // 1.5 GiB: six Infinity Caches
    char* buf; float* out;
    CK(hipMalloc(&buf, span)); CK(hipMalloc(&out, 4096)); CK(hipMemset(buf, 0, span));
    CK(hipDeviceSynchronize());
    auto rd = [&](const char* name, auto kern, int bytes) {
        const unsigned row_bytes = 256u * bytes;
        const unsigned rows = (unsigned)(((size_t)1 << 30) / row_bytes);        // 1 GiB moved
        hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, (const char*)buf, out, rows, row_bytes);
        CK(hipDeviceSynchronize());
        printf("%s bytes %zu\n", name, (size_t)rows * row_bytes);
    };
    auto wr = [&](const char* name, auto kern, int bytes) {
        const unsigned row_bytes = 256u * bytes;
        const unsigned rows = (unsigned)(((size_t)1 << 30) / row_bytes);
        hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, buf, rows, row_bytes);
        CK(hipDeviceSynchronize());
        printf("%s bytes %zu\n", name, (size_t)rows * row_bytes);
    };
    // between two kernels the previous one's 1 GiB has pushed everything older out of the Infinity Cache
    rd("k_read<4>", k_read<4>, 4);
    rd("k_read<12>", k_read<12>, 12);
    rd("k_read<16>", k_read<16>, 16);
    rd("k_read<12>", k_read<12>, 12);
    wr("k_write<12>", k_write<12>, 12);
    wr("k_write<16>", k_write<16>, 16);
    return 0;
}
