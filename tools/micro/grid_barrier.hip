// What does a stage boundary cost on gfx950 -- a kernel boundary against a grid-wide barrier inside ONE launch?
// The small-grid DP plan (201x37x20 x 10 stages) is nine dependent sweeps of ~10 us, each as long as one workgroup's
// chain (docs/NOTEBOOK.md 2.2d); this probe runs the SHAPE of such a plan -- 740 workgroups of 256 lanes, every stage reads
// four 12-byte rows that OTHER workgroups (other XCDs) wrote in the stage before, a little arithmetic, one 12-byte
// store per lane -- in four forms and checks the result of each against the host:
//   0  one launch per stage                                     (what the planner does today)
//   1  one launch, barrier = ONE device counter                 (release add, relaxed poll, acquire fence)
//   2  one launch, barrier = a counter per XCD + a device one   (the last arrival of an XCD forwards)
//   3  hipLaunchCooperativeKernel + cooperative_groups grid sync
//   5  form 0 captured once into a hipGraph, one hipGraphLaunch per run
//   4  as 2, but only the LAST arrival of an XCD releases (one L2 write-back per XCD and stage): the others wait for their
//      stores (vmcnt(0): a store is acknowledged by the L2) and add to the XCD's counter without a fence.  Relies on
//      blockIdx.x & 7 being the XCD of a workgroup -- a property of the dispatcher, not of the memory model
// and times launch -> completion on the host for a 1-stage and a 10-stage run of each.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier tools/micro/grid_barrier.hip && ./grid_barrier [workgroups] [stages]
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
namespace cg = cooperative_groups;

typedef float v3f __attribute__((ext_vector_type(3)));
constexpr int LANES = 256;
constexpr unsigned ABORT = 0x80000000u;
constexpr int SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ v3f stage_body(const char* prev, int wg, int n_wg, int lane)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(prev), 0, n_wg * LANES * 12, 0x00020000);
    v3f acc = {1.0f, 2.0f, 3.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = (wg + 1 + 37 * k) % n_wg;                       // rows of other workgroups, other XCDs
        const v3f v = __builtin_bit_cast(v3f, __builtin_amdgcn_raw_buffer_load_b96(r, lane * 12, row * LANES * 12, 0));
        acc += v * 0.25f;
    }
    return acc;
}

__device__ __forceinline__ void store_row(char* cur, int wg, int lane, v3f v)
{
    *reinterpret_cast<v3f*>(cur + ((size_t)wg * LANES + lane) * 12) = v;
}

__global__ void __launch_bounds__(LANES) k_stage(const char* prev, char* cur, int n_wg)
{
    store_row(cur, blockIdx.x, threadIdx.x, stage_body(prev, blockIdx.x, n_wg, threadIdx.x));
}

// arrive + wait; returns false when the wait was given up (the counter then carries ABORT and every waiter leaves)
__device__ __forceinline__ bool barrier_flat(unsigned* ctr, unsigned target)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        unsigned v;
        while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            if (++spins > SPIN_LIMIT) { __hip_atomic_fetch_or(ctr, ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = !(v & ABORT) && spins <= SPIN_LIMIT;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = ok;
    __syncthreads();
    return s_ok;
}

__device__ __forceinline__ bool barrier_xcd(unsigned* xcd_ctr, unsigned* ctr, int per_xcd, unsigned stage)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        // the last arrival of this XCD's share forwards ONE add to the device counter
        const unsigned mine = __hip_atomic_fetch_add(xcd_ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (mine + 1 == (stage + 1) * (unsigned)per_xcd) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (stage + 1) * 8u;
        int spins = 0;
        unsigned v;
        while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            if (++spins > SPIN_LIMIT) { __hip_atomic_fetch_or(ctr, ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = !(v & ABORT) && spins <= SPIN_LIMIT;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = ok;
    __syncthreads();
    return s_ok;
}

__device__ __forceinline__ bool barrier_xcd_last(unsigned* xcd_ctr, unsigned* ctr, int per_xcd, unsigned stage)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned mine = __hip_atomic_fetch_add(xcd_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mine + 1 == (stage + 1) * (unsigned)per_xcd) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (stage + 1) * 8u;
        int spins = 0;
        unsigned v;
        while ((v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            if (++spins > SPIN_LIMIT) { __hip_atomic_fetch_or(ctr, ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = !(v & ABORT) && spins <= SPIN_LIMIT;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = ok;
    __syncthreads();
    return s_ok;
}

template <int FORM>
__global__ void __launch_bounds__(LANES) k_plan(char* t0, char* t1, int n_wg, int stages, unsigned* ctr, int* failed)
{
    char* tab[2] = {t0, t1};
    const int wg = blockIdx.x, lane = threadIdx.x;
    for (int st = 0; st < stages; ++st) {
        store_row(tab[(st + 1) & 1], wg, lane, stage_body(tab[st & 1], wg, n_wg, lane));
        if (st + 1 == stages) break;
        bool ok = true;
        if constexpr (FORM == 1) ok = barrier_flat(ctr, (unsigned)(st + 1) * (unsigned)n_wg);
        if constexpr (FORM == 2) {
            const int x = wg & 7;
            const int per_xcd = (n_wg - x + 7) / 8;
            ok = barrier_xcd(ctr + 16 * (1 + x), ctr, per_xcd, (unsigned)st);
        }
        if constexpr (FORM == 4) {
            const int x = wg & 7;
            ok = barrier_xcd_last(ctr + 16 * (1 + x), ctr, (n_wg - x + 7) / 8, (unsigned)st);
        }
        if constexpr (FORM == 3) cg::this_grid().sync();
        if (!ok) { if (lane == 0) *failed = 1; return; }
    }
}

static void host_reference(std::vector<float>& tab, int n_wg, int stages)
{
    std::vector<float> cur(tab.size());
    for (int st = 0; st < stages; ++st) {
        for (int wg = 0; wg < n_wg; ++wg)
            for (int lane = 0; lane < LANES; ++lane) {
                float acc[3] = {1.0f, 2.0f, 3.0f};
                for (int k = 0; k < 4; ++k) {
                    const int row = (wg + 1 + 37 * k) % n_wg;
                    for (int c = 0; c < 3; ++c) acc[c] += tab[((size_t)row * LANES + lane) * 3 + c] * 0.25f;
                }
                for (int c = 0; c < 3; ++c) cur[((size_t)wg * LANES + lane) * 3 + c] = acc[c];
            }
        tab.swap(cur);
    }
}

int main(int argc, char** argv)
{
    const int n_wg = argc > 1 ? atoi(argv[1]) : 740;
    const int stages = argc > 2 ? atoi(argv[2]) : 10;
    const size_t bytes = (size_t)n_wg * LANES * 12;
    std::vector<float> init(bytes / 4);
    for (size_t i = 0; i < init.size(); ++i) init[i] = (float)(i % 977) * 0.001f;
    std::vector<float> want = init;
    host_reference(want, n_wg, stages);
    char *t0, *t1;
    unsigned* ctr;
    int* failed;
    CK(hipMalloc(&t0, bytes)); CK(hipMalloc(&t1, bytes)); CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&failed, 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_plan<1>, LANES, 0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%d workgroups of %d lanes, %d stages; co-resident capacity %d x %d CUs = %d\n", n_wg, LANES, stages, occ, prop.multiProcessorCount,
           occ * prop.multiProcessorCount);
    if (n_wg > occ * prop.multiProcessorCount) { printf("grid does not fit: persistent forms would deadlock\n"); return 1; }
    std::vector<float> got(init.size());
    for (int form = 0; form < 6; ++form) {
        for (int n_st : {1, stages}) {
            double best = 1e30, sum = 0.0;
            const int reps = 200;
            bool good = true;
            hipGraphExec_t gexec = nullptr;
            if (form == 5) {
                hipGraph_t graph;
                char* tab[2] = {t0, t1};
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int st = 0; st < n_st; ++st) hipLaunchKernelGGL(k_stage, dim3(n_wg), dim3(LANES), 0, s, tab[st & 1], tab[(st + 1) & 1], n_wg);
                CK(hipStreamEndCapture(s, &graph));
                CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
                CK(hipGraphDestroy(graph));
            }
            for (int rep = 0; rep < reps + 20; ++rep) {
                CK(hipMemcpyAsync(t0, init.data(), bytes, hipMemcpyHostToDevice, s));
                CK(hipMemsetAsync(ctr, 0, 4096, s));
                CK(hipMemsetAsync(failed, 0, 4, s));
                CK(hipStreamSynchronize(s));
                const auto a = std::chrono::steady_clock::now();
                if (form == 0) {
                    char* tab[2] = {t0, t1};
                    for (int st = 0; st < n_st; ++st) hipLaunchKernelGGL(k_stage, dim3(n_wg), dim3(LANES), 0, s, tab[st & 1], tab[(st + 1) & 1], n_wg);
                } else if (form == 1) {
                    hipLaunchKernelGGL(k_plan<1>, dim3(n_wg), dim3(LANES), 0, s, t0, t1, n_wg, n_st, ctr, failed);
                } else if (form == 2) {
                    hipLaunchKernelGGL(k_plan<2>, dim3(n_wg), dim3(LANES), 0, s, t0, t1, n_wg, n_st, ctr, failed);
                } else if (form == 5) {
                    CK(hipGraphLaunch(gexec, s));
                } else if (form == 4) {
                    hipLaunchKernelGGL(k_plan<4>, dim3(n_wg), dim3(LANES), 0, s, t0, t1, n_wg, n_st, ctr, failed);
                } else {
                    int nw = n_wg, ns = n_st;
                    void* args[] = {&t0, &t1, &nw, &ns, &ctr, &failed};
                    CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(k_plan<3>), dim3(n_wg), dim3(LANES), args, 0, s));
                }
                CK(hipStreamSynchronize(s));
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
                if (rep >= 20) { best = us < best ? us : best; sum += us; }
                if (rep == reps + 19 && n_st == stages) {
                    int f = 0;
                    CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(got.data(), (n_st & 1) ? t1 : t0, bytes, hipMemcpyDeviceToHost));
                    size_t bad = 0;
                    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
                    good = !f && bad == 0;
                    if (!good) printf("   form %d: failed flag %d, %zu of %zu values differ\n", form, f, bad, got.size());
                }
            }
            if (gexec) CK(hipGraphExecDestroy(gexec));
            printf("form %d  %2d stage(s): launch -> completion %7.2f us mean, %7.2f us min%s\n", form, n_st, sum / reps, best,
                   n_st == stages ? (good ? "   result == host" : "   RESULT WRONG") : "");
        }
    }
    return 0;
}
