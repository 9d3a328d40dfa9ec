// mfma_f64_probe: would v_mfma_f64_4x4x4 shorten the backward Riccati stage of the batched iLQR kernel?
//
// One stage of riccati_backward (tpl_amd/csrc/ilqr/ilqr_model.hip; X = 7, U = 2) is two small products on ONE wavefront
// per SIMD, each behind an LDS exchange, on a chain that is strictly sequential over the 50 stages:
//     phase A   W = Vxx [fx fu]          7x7 . 7x9        (lane = entry of W, 7 LDS operand pairs, 7 dependent FMAs)
//     phase B   Q = L + [fx fu]^T W      9x7 . 7x10       (lane = entry of Q, the same)
// This probe times exactly that chain -- LDS write, wait, operand reads, arithmetic, LDS write, wait -- in two forms,
// with a dependent hand-over between the stages (the next Vxx is formed from Q), on one wavefront per SIMD of every CU:
//     fma    the kernel's form: every lane reads its 7 + 7 operands from LDS (dot_lds), 7 dependent v_fma_f64
//     mfma   the products padded to 8x8 . 8x12 and 12x8 . 8x12, as 4x4 blocks on v_mfma_f64_4x4x4_4b (four independent
//            4x4x4 block products per instruction, one operand double per lane): K = 8 is two dependent MFMAs per
//            output block; A: 6 blocks -> 2 + 2 instructions, B: 9 blocks -> 3 + 3; operands come from LDS in the
//            instruction's lane layout (one ds_read_b64 per operand per instruction), results go back through LDS
// and checks the two against each other (same sums in a different order: agreement to ~1e-13 relative).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe tools/micro/mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NX = 7, NU = 2, NXU = 9, T = 50;
constexpr int PA = 8, PB = 12;       // padded: Vxx 8x8, F 8x12 (columns 9..11 zero), W 8x12, Q 12x12

struct Lds {
    double Vxx[PA * PA];             // row-major 8x8 (row / column 7 zero)
    double F[PA * PB];               // [k][j]: 8 x 12
    double W[PA * PB];               // [i][j]: 8 x 12
    double Q[PB * PB];               // 12 x 12
};

__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the stage hand-over both forms share: next Vxx = 0.5 * (Qxx + Qxx^T) * 0.125 + I  (keeps the chain dependent and bounded)
__device__ __forceinline__ void next_vxx(Lds* s, int lane)
{
    const int i = lane >> 3, j = lane & 7;
    double v = 0.0;
    if (i < NX && j < NX) v = 0.0625 * (s->Q[i * PB + j] + s->Q[j * PB + i]) + (i == j ? 1.0 : 0.0);
    s->Vxx[i * PA + j] = v;
}

__global__ void __launch_bounds__(64) k_fma(const double* __restrict__ Fall, double* __restrict__ out, long long* __restrict__ cyc)
{
    __shared__ Lds s;
    const int lane = threadIdx.x;
    for (int e = lane; e < PA * PA; e += 64) s.Vxx[e] = ((e >> 3) == (e & 7) && (e >> 3) < NX) ? 1.0 : 0.0;
    lds_sync();
    const long long t0 = __builtin_amdgcn_s_memtime();
    double f0 = Fall[lane], f1 = Fall[min(lane + 64, PA * PB - 1)];          // stage 0; the next stage is requested a stage ahead, as in the solver
    for (int t = 0; t < T; ++t) {
        s.F[lane] = f0; if (lane + 64 < PA * PB) s.F[lane + 64] = f1;
        const double* Fn = Fall + (size_t)min(t + 1, T - 1) * PA * PB;
        f0 = Fn[lane]; f1 = Fn[min(lane + 64, PA * PB - 1)];
        lds_sync();
        {   // phase A: lane = (i, j) of W, 63 entries
            const int e = min(lane, NX * NXU - 1), i = e / NXU, j = e - i * NXU;
            double a[NX], b[NX];
#pragma unroll
            for (int k = 0; k < NX; ++k) { a[k] = s.Vxx[i * PA + k]; b[k] = s.F[k * PB + j]; }
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NX; ++k) acc = fma(a[k], b[k], acc);
            s.W[i * PB + j] = acc;
        }
        lds_sync();
        {   // phase B: Q[c][d] = sum_k F[k][c] W[k][d], c < 9, d < 9 (81 entries on 64 lanes: two passes, as E_B = 1 holds only with the
            // packed triangle; the full square costs the probe's FMA form one extra pass, which favours the MFMA form)
            for (int e = lane; e < NXU * NXU; e += 64) {
                const int c = e / NXU, d = e - c * NXU;
                double a[NX], b[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) { a[k] = s.F[k * PB + c]; b[k] = s.W[k * PB + d]; }
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NX; ++k) acc = fma(a[k], b[k], acc);
                s.Q[c * PB + d] = acc;
            }
        }
        lds_sync();
        next_vxx(&s, lane);
        lds_sync();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) out[lane] = s.Vxx[lane];
}

// v_mfma_f64_4x4x4_4b: every lane holds ONE double of A, of B and of D.  The lane number has three 2-bit fields
// (l & 3, (l >> 2) & 3, l >> 4); which field is the row / column, the k index and the block of each fragment is found at
// run time: main() tries the 6 x 6 x 6 assignments and keeps the one that reproduces the FMA form.
// pa: A fragment (i, k, block) = the three fields in the order of permutation pa; pb: B (j, k, block); pd: D (j, i, block)
__device__ __forceinline__ void fields(int lane, int perm, int& x, int& y, int& z)
{
    const int f[3] = {lane & 3, (lane >> 2) & 3, lane >> 4};
    const int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    x = f[P[perm][0]]; y = f[P[perm][1]]; z = f[P[perm][2]];
}

__global__ void __launch_bounds__(64) k_mfma(const double* __restrict__ Fall, double* __restrict__ out, long long* __restrict__ cyc,
                                             int pa, int pb, int pd)
{
    __shared__ Lds s;
    const int lane = threadIdx.x;
    int ai, ak, ab, bj, bk, bb, dj, di, db;
    fields(lane, pa, ai, ak, ab);
    fields(lane, pb, bj, bk, bb);
    fields(lane, pd, dj, di, db);
    for (int e = lane; e < PA * PA; e += 64) s.Vxx[e] = ((e >> 3) == (e & 7) && (e >> 3) < NX) ? 1.0 : 0.0;
    lds_sync();
    const long long t0 = __builtin_amdgcn_s_memtime();
    double f0 = Fall[lane], f1 = Fall[min(lane + 64, PA * PB - 1)];
    for (int t = 0; t < T; ++t) {
        s.F[lane] = f0; if (lane + 64 < PA * PB) s.F[lane + 64] = f1;
        const double* Fn = Fall + (size_t)min(t + 1, T - 1) * PA * PB;
        f0 = Fn[lane]; f1 = Fn[min(lane + 64, PA * PB - 1)];
        lds_sync();
        // phase A: W (8 x 12) = Vxx (8 x 8) F (8 x 12): output blocks (row block, column block) = (blk / 3, blk % 3), six of them:
        // instruction 0 carries blocks 0..3, instruction 1 blocks 4, 5 (and two repeats)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int blkA = min(g * 4 + ab, 5), blkB = min(g * 4 + bb, 5), blkD = g * 4 + db;
            double acc = 0.0;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const double a = s.Vxx[((blkA / 3) * 4 + ai) * PA + kk * 4 + ak];
                const double bv = s.F[(kk * 4 + bk) * PB + (blkB % 3) * 4 + bj];
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bv, acc, 0, 0, 0);
            }
            if (blkD < 6) s.W[((blkD / 3) * 4 + di) * PB + (blkD % 3) * 4 + dj] = acc;
        }
        lds_sync();
        // phase B: Q (12 x 12) = F^T (12 x 8) W (8 x 12): nine output blocks, three instructions
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int blkA = min(g * 4 + ab, 8), blkB = min(g * 4 + bb, 8), blkD = g * 4 + db;
            double acc = 0.0;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const double a = s.F[(kk * 4 + ak) * PB + (blkA / 3) * 4 + ai];        // A = F^T
                const double bv = s.W[(kk * 4 + bk) * PB + (blkB % 3) * 4 + bj];
                acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bv, acc, 0, 0, 0);
            }
            if (blkD < 9) s.Q[((blkD / 3) * 4 + di) * PB + (blkD % 3) * 4 + dj] = acc;
        }
        lds_sync();
        next_vxx(&s, lane);
        lds_sync();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) out[lane] = s.Vxx[lane];
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4;              // one wavefront per SIMD, as the solver runs
    std::vector<double> F((size_t)T * PA * PB, 0.0);
    srand(7);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < NX; ++k)
            for (int j = 0; j < NXU; ++j) F[(size_t)t * PA * PB + k * PB + j] = (k == j ? 1.0 : 0.0) + 0.1 * ((double)rand() / RAND_MAX - 0.5);
    double *dF, *dout; long long* dcyc;
    CK(hipMalloc(&dF, F.size() * 8)); CK(hipMalloc(&dout, 64 * 8)); CK(hipMalloc(&dcyc, blocks * 8));
    CK(hipMemcpy(dF, F.data(), F.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> res[2];
    const char* names[2] = {"fma (dot_lds form)", "mfma_f64_4x4x4_4b"};
    auto run = [&](int form, int la, int lb, int ld, bool print, int nblk) {
        std::vector<long long> cyc(blocks);
        for (int rep = 0; rep < 3; ++rep) {
            if (form == 0) hipLaunchKernelGGL(k_fma, dim3(nblk), dim3(64), 0, 0, dF, dout, dcyc);
            else hipLaunchKernelGGL(k_mfma, dim3(nblk), dim3(64), 0, 0, dF, dout, dcyc, la, lb, ld);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(cyc.data(), dcyc, nblk * 8, hipMemcpyDeviceToHost));
        res[form].resize(64);
        CK(hipMemcpy(res[form].data(), dout, 64 * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (int i = 0; i < nblk; ++i) sum += (double)cyc[i];
        if (print) printf("%-22s %8.1f shader cycles per stage (s_memtime, mean over %d wavefronts, %d stages)\n", names[form], sum / blocks / T, blocks, T);
    };
    auto diff = [&]() {
        double w = 0.0;
        for (int i = 0; i < 64; ++i) w = std::fmax(w, std::fabs(res[0][i] - res[1][i]) / std::fmax(1.0, std::fabs(res[0][i])));
        return w;
    };
    run(0, 0, 0, 0, true, blocks);
    int best = -1;
    for (int m = 0; m < 216 && best < 0; ++m) {
        run(1, m % 6, (m / 6) % 6, m / 36, false, 1);
        if (diff() < 1e-9) best = m;
    }
    if (best < 0) { printf("no fragment layout reproduces the FMA form\n"); return 1; }
    const char* P[6] = {"(f0, f1, f2)", "(f0, f2, f1)", "(f1, f0, f2)", "(f1, f2, f0)", "(f2, f0, f1)", "(f2, f1, f0)"};
    printf("fragment layout, lane fields f0 = l & 3, f1 = (l >> 2) & 3, f2 = l >> 4:  A (i, k, block) = %s, B (j, k, block) = %s, D (j, i, block) = %s\n",
           P[best % 6], P[(best / 6) % 6], P[best / 36]);
    run(1, best % 6, (best / 6) % 6, best / 36, true, blocks);
    printf("largest relative difference between the two forms' final Vxx: %.3e (same product, sums in a different order)\n", diff());
    return 0;
}
