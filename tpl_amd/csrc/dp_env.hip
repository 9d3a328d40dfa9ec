// dp_env.hip -- SURVEY 8(f-1): the occupancy pipeline that feeds the DP planner, for gfx950.
//
// Replaces, behind the same DynProgEnvironment surface (bindings_dp_env.cu:53-105):
//   insertGeometry / setRefLine / update            env.cu:352-437
//   the OpenGL occupancy renderer                   occupancy_renderer.cu:275-320
//   occupancyToFrenet / closeIntersections / updateDistMapLon      env.cu:11-129
//
// Re-design, not a translation:
//  * no OpenGL/EGL context, frame buffer object or GL<->compute interop: the triangle soup is
//    rasterised by a compute kernel (one wavefront per triangle, exact integer edge functions
//    on vertices snapped to 1/256 pixel, OR-blend = atomicOr), straight into the uint32 image
//    the next kernel samples;
//  * the dilation disc of occupancyToFrenet is sampled ONCE per Frenet cell (s, l) and the
//    resulting time-bit word is expanded to all t slices (the reference repeats the 144 point
//    samples and the reference-line interpolation with its sin/cos for every slice);
//  * the reference-line frame of a station (position, sin/cos of the heading, lane bounds, speed
//    limit) is computed once per station on the host in the reference's own float/double mix --
//    ~200 values per update -- instead of once per cell on the device;
//  * all stages are enqueued on one stream with a single synchronisation at the end (the
//    reference synchronises the device after each of its three kernels).
//
// Rasterisation rule (documented because OpenGL leaves it to the implementation; the CPU oracle
// oracle/dp_env_oracle.c states the same rule and the two are compared bit for bit):
//   window x = (x * scale) * (W/2) + W/2 in float (vertex shader, then viewport transform),
//   snapped to 8 sub-pixel bits by round-to-nearest-even; a pixel is covered when its centre
//   (i + 1/2, j + 1/2) is inside the triangle, pixels exactly on an edge belong to the triangle
//   iff the edge is a left edge or a horizontal top edge (y up); zero-area triangles produce
//   nothing; both windings are drawn (no culling in the reference).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dp_env_internal.h"
#include "dp_lon_math.h"

using namespace tplhip;

namespace {

constexpr float PPM = 4.0f;          // pixels per metre, env.cu:395
constexpr int SUB = 256;             // sub-pixel grid

struct SnapVertex { int x, y; };

__host__ __device__ inline SnapVertex snap_vertex(float vx, float vy, float scale, float half_w)
{
    // vertex shader: clip = v * scale (occupancy_renderer.cu:22-27); viewport: w = ndc * W/2 + W/2
    const float cx = vx * scale, cy = vy * scale;
    float wx = cx * half_w + half_w, wy = cy * half_w + half_w;
    // keep far-away vertices inside the integer range of the edge functions (1024 km at 4 px/m)
    const float lim = 4194304.0f;
    wx = fminf(fmaxf(wx, -lim), lim);
    wy = fminf(fmaxf(wy, -lim), lim);
    SnapVertex s;
    s.x = (int)rintf(wx * (float)SUB);
    s.y = (int)rintf(wy * (float)SUB);
    return s;
}

__host__ __device__ inline long long edge_fn(const SnapVertex& a, const SnapVertex& b, long long px, long long py)
{
    return ((long long)b.x - a.x) * (py - a.y) - ((long long)b.y - a.y) * (px - a.x);
}

// pixels exactly on edge a->b of a counter-clockwise (y up) triangle belong to it iff the edge is
// a left edge (going down) or a horizontal top edge (going left)
__host__ __device__ inline bool edge_owns(const SnapVertex& a, const SnapVertex& b)
{
    const long long dx = (long long)b.x - a.x, dy = (long long)b.y - a.y;
    return dy < 0 || (dy == 0 && dx < 0);
}

// ------------------------------------------------------------------------
// rasteriser: one wavefront per triangle, OR of the vertex flags into the image.
// The fragment colour is the flat-shaded flags of the provoking (last) vertex,
// occupancy_renderer.cu:17-37 ("flat out uint color").
// ------------------------------------------------------------------------
__global__ void k_raster_or(uint32_t* __restrict__ img, int W, const EnvVertex* __restrict__ vtx, int n_tri,
                            float scale)
{
    const int tri = blockIdx.x;
    if (tri >= n_tri) return;
    const float half_w = (float)W * 0.5f;
    const EnvVertex e0 = vtx[3 * tri], e1 = vtx[3 * tri + 1], e2 = vtx[3 * tri + 2];
    SnapVertex v0 = snap_vertex(e0.x, e0.y, scale, half_w);
    SnapVertex v1 = snap_vertex(e1.x, e1.y, scale, half_w);
    SnapVertex v2 = snap_vertex(e2.x, e2.y, scale, half_w);
    const uint32_t colour = e2.flags;
    long long area = edge_fn(v0, v1, v2.x, v2.y);
    if (area == 0) return;
    if (area < 0) { const SnapVertex t = v1; v1 = v2; v2 = t; }
    const int min_x = min(v0.x, min(v1.x, v2.x)), max_x = max(v0.x, max(v1.x, v2.x));
    const int min_y = min(v0.y, min(v1.y, v2.y)), max_y = max(v0.y, max(v1.y, v2.y));
    // pixel (i, j) has its centre at (i*256 + 128, j*256 + 128)
    const int i_lo = max(0, (min_x - SUB / 2) >> 8), i_hi = min(W - 1, (max_x - SUB / 2 + SUB - 1) >> 8);
    const int j_lo = max(0, (min_y - SUB / 2) >> 8), j_hi = min(W - 1, (max_y - SUB / 2 + SUB - 1) >> 8);
    if (i_lo > i_hi || j_lo > j_hi) return;
    const bool own01 = edge_owns(v0, v1), own12 = edge_owns(v1, v2), own20 = edge_owns(v2, v0);
    const int bw = i_hi - i_lo + 1;
    const long long n_px = (long long)bw * (j_hi - j_lo + 1);
    for (long long q = threadIdx.x; q < n_px; q += blockDim.x) {
        const int j = j_lo + (int)(q / bw);
        const int i = i_lo + (int)(q - (long long)(j - j_lo) * bw);
        const long long px = (long long)i * SUB + SUB / 2, py = (long long)j * SUB + SUB / 2;
        const long long w0 = edge_fn(v0, v1, px, py), w1 = edge_fn(v1, v2, px, py), w2 = edge_fn(v2, v0, px, py);
        const bool in = (w0 > 0 || (w0 == 0 && own01)) && (w1 > 0 || (w1 == 0 && own12))
                        && (w2 > 0 || (w2 == 0 && own20));
        if (in) atomicOr(&img[(size_t)j * W + i], colour);
    }
}

// tex2D<unsigned int>, point filter, unnormalised coordinates, clamp addressing
// (occupancy_renderer.cu:339-346): texel = floor(coordinate), clamped
__device__ __forceinline__ uint32_t fetch_px(const uint32_t* __restrict__ img, int W, float x, float y)
{
    const float fx = fminf(fmaxf(floorf(x), 0.0f), (float)(W - 1));
    const float fy = fminf(fmaxf(floorf(y), 0.0f), (float)(W - 1));
    return img[(size_t)(int)fy * W + (int)fx];
}

// ------------------------------------------------------------------------
// occupancyToFrenet, env.cu:11-63.  One workgroup per station: the (lateral cell, disc sample)
// pairs -- 31 x 144 point samples for the default parameters -- are spread over its threads and
// OR-ed into one word per cell in LDS (the reference walks the 144 samples of a cell one after
// the other, and once per time slice); the time-bit word of a cell is then expanded to all
// slices by all threads.  occ: [l][s][t], t fastest.
// ------------------------------------------------------------------------
constexpr int OTF_MAX_L = 1024;

__global__ void k_occ_to_frenet(float* __restrict__ occ, const uint32_t* __restrict__ img, int W,
                                const StationFrame* __restrict__ frames, TplDpEnvParams ep)
{
    __shared__ uint32_t sh_bits[OTF_MAX_L];
    __shared__ float sh_px[OTF_MAX_L], sh_py[OTF_MAX_L];
    const int idx_s = blockIdx.x;
    const int L = ep.l_steps, T = ep.t_steps;
    const int dil_steps = (int)ceilf(ep.dilation * PPM);
    const StationFrame f = frames[idx_s];
    for (int idx_l = threadIdx.x; idx_l < L; idx_l += blockDim.x) {
        const float l = ep.l_min + (float)idx_l * ep._l_step_size;
        const float x = (float)(f.x - (double)l * f.sin_h);
        const float y = (float)(f.y + (double)l * f.cos_h);
        sh_px[idx_l] = (float)((double)(x * PPM) + (double)(float)W / 2.0);
        sh_py[idx_l] = (float)((double)(y * PPM) + (double)(float)W / 2.0);
        sh_bits[idx_l] = 0u;
    }
    __syncthreads();
    const int side = 2 * dil_steps, nq = side * side;
    for (int w = threadIdx.x; w < L * nq; w += blockDim.x) {
        const int idx_l = w / nq, q = w - idx_l * nq;
        const int i = q / side - dil_steps, j = q - (q / side) * side - dil_steps;
        if (sqrtf((float)(i * i + j * j)) > (float)dil_steps) continue;
        const uint32_t px = fetch_px(img, W, sh_px[idx_l] + (float)i, sh_py[idx_l] + (float)j);
        if (px) atomicOr(&sh_bits[idx_l], px);
    }
    __syncthreads();
    for (int w = threadIdx.x; w < L * T; w += blockDim.x) {
        const int idx_l = w / T, t = w - idx_l * T;
        const float l = ep.l_min + (float)idx_l * ep._l_step_size;
        const uint32_t bits = sh_bits[idx_l];
        const bool stationary = (bits >> 31) != 0u;
        const bool outside = ((double)l > f.d_left || (double)l < -f.d_right)
                             || (idx_l == 0 || idx_l == L - 1)
                             || (f.v_max < 0.1);
        // (occ & (1 << t)): a shift by >= 32 yields 0 on the reference's hardware
        float val = (t < 32 && ((bits >> t) & 1u)) ? 1.0f : 0.0f;
        if (stationary) val = 3.0f;
        if (outside) val = 2.0f;
        occ[((size_t)idx_l * ep.s_steps + idx_s) * T + t] = val;
    }
}

// ------------------------------------------------------------------------
// closeIntersections, env.cu:65-93.  One thread per (t, s).
// ------------------------------------------------------------------------
__global__ void k_close_intersections(float* __restrict__ occ, const float* __restrict__ semantic, int ref_steps,
                                      float ref_step, TplDpEnvParams ep)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ep.t_steps * ep.s_steps) return;
    const int idx_s = idx / ep.t_steps;
    const int idx_t = idx - idx_s * ep.t_steps;
    const float s = ep.s_min + (float)idx_s * ep._s_step_size;
    // texLerp(s).w: hardware-filtered lookup in the reference line texture (common.cuh:93-95),
    // filter rule R2/R3 of DESIGN.md
    const Tap tp = make_tap(s / ref_step + 0.5f, ref_steps);
    const float sem = lerp1(semantic[tp.i0], semantic[tp.i1], tp.w);
    if (sem < 0.5f) return;
    bool close = false;
    for (int idx_l = 1; idx_l < ep.l_steps - 1; ++idx_l) {
        if (occ[((size_t)idx_l * ep.s_steps + idx_s) * ep.t_steps + idx_t] == 1.0f) close = true;
    }
    if (close) {
        for (int idx_l = 1; idx_l < ep.l_steps - 1; ++idx_l)
            occ[((size_t)idx_l * ep.s_steps + idx_s) * ep.t_steps + idx_t] = 1.0f;
    }
}

// ------------------------------------------------------------------------
// updateDistMapLon, env.cu:95-129.  The two scans along s are order-dependent float
// accumulations, one chain per (l, t).  A workgroup takes one l and a block of DML_TC time
// slices: the occupancy tile comes into LDS in one coalesced pass, the chains run from and to LDS
// (a chain that loads and stores global memory in turn pays a write acknowledgement per station),
// and the result leaves in one coalesced pass.  dist_lon: [l][s][t][2].
// ------------------------------------------------------------------------
constexpr int DML_TC = 16;                            // time slices per workgroup where LDS allows

__global__ void k_dist_map_lon(float* __restrict__ dist_lon, const float* __restrict__ occ, TplDpEnvParams ep, int tc_cap)
{
    extern __shared__ __attribute__((aligned(16))) float dml_sh[];
    const int S = ep.s_steps, T = ep.t_steps;
    const int idx_l = blockIdx.x, t0 = blockIdx.y * tc_cap;
    const int tc = min(tc_cap, T - t0);
    float* sh_o = dml_sh;                               // [S][tc]
    float* sh_d = dml_sh + S * tc_cap;                  // [S][tc][2]
    const float* o = occ + (size_t)idx_l * S * T + t0;
    float* d2 = dist_lon + ((size_t)idx_l * S * T + t0) * 2;
    for (int e = threadIdx.x; e < S * tc; e += blockDim.x) {
        const int is = e / tc, tt = e - is * tc;
        sh_o[e] = o[(size_t)is * T + tt];
    }
    __syncthreads();
    // one thread per (slice, direction): the towards-the-ego and away-from-the-ego chains of a slice
    // are independent.  Eight occupancy flags are read before the chain runs over them (the reads
    // do not depend on the running distance; one at a time each step would wait for its LDS read).
    for (int w = threadIdx.x; w < 2 * tc; w += blockDim.x) {
        const int tt = w % tc;
        const bool fwd = w >= tc;
        float d = 10000.0f;
        int k0 = 0;
        for (; k0 + 8 <= S; k0 += 8) {
            float o8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx_s = fwd ? k0 + k : S - 1 - (k0 + k);
                o8[k] = sh_o[idx_s * tc + tt];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx_s = fwd ? k0 + k : S - 1 - (k0 + k);
                if (o8[k] > 0.0f) d = 0.0f; else d += ep._s_step_size;
                sh_d[(idx_s * tc + tt) * 2 + (fwd ? 1 : 0)] = d;
            }
        }
        for (; k0 < S; ++k0) {
            const int idx_s = fwd ? k0 : S - 1 - k0;
            if (sh_o[idx_s * tc + tt] > 0.0f) d = 0.0f; else d += ep._s_step_size;
            sh_d[(idx_s * tc + tt) * 2 + (fwd ? 1 : 0)] = d;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * tc * 2; e += blockDim.x) {
        const int is = e / (tc * 2), r = e - is * (tc * 2);
        d2[(size_t)is * T * 2 + r] = sh_d[e];
    }
}

// interpDistField, env.cu:225-234, at one point (DynProgEnvironment::getDist)
__global__ void k_get_dist(float* __restrict__ out, const float* __restrict__ occ, float t, float s, float l,
                           TplDpEnvParams ep)
{
    const float t_idx = env_t_idx(ep, t);
    const float sc = (s - ep.s_min) / (ep.s_max - ep.s_min) * (float)(ep.s_steps - 1);
    const float lc = (l - ep.l_min) / (ep.l_max - ep.l_min) * (float)(ep.l_steps - 1);
    const Tap tt = make_tap(t_idx + 0.5f, ep.t_steps);
    const Tap ts = make_tap(sc + 0.5f, ep.s_steps);
    const Tap tl = make_tap(lc + 0.5f, ep.l_steps);
    auto at = [&](int il, int is, int it) { return occ[((size_t)il * ep.s_steps + is) * ep.t_steps + it]; };
    // rule R3: z (l) first, then y (s), then x (t)
    const float b00 = lerp1(at(tl.i0, ts.i0, tt.i0), at(tl.i1, ts.i0, tt.i0), tl.w);
    const float b10 = lerp1(at(tl.i0, ts.i1, tt.i0), at(tl.i1, ts.i1, tt.i0), tl.w);
    const float b01 = lerp1(at(tl.i0, ts.i0, tt.i1), at(tl.i1, ts.i0, tt.i1), tl.w);
    const float b11 = lerp1(at(tl.i0, ts.i1, tt.i1), at(tl.i1, ts.i1, tt.i1), tl.w);
    out[0] = lerp1(lerp1(b00, b10, ts.w), lerp1(b01, b11, ts.w), tt.w);
}

template <typename T>
int ensure_capacity(T** ptr, size_t* cap, size_t need, const char* what)
{
    if (need <= *cap && *ptr) return TPLHIP_OK;
    (void)hipFree(*ptr);
    *ptr = nullptr; *cap = 0;
    const size_t n = std::max<size_t>(need, 1);
    HIP_TRY(hipMalloc(ptr, n * sizeof(T)), what);
    *cap = n;
    return TPLHIP_OK;
}

} // namespace

extern "C" {

int tplhip_dp_env_set_ref_line(tplhip_dp_env* env, const double* ref_line, int n, float step_size)
{
    if (!env || (!ref_line && n > 0) || n < 0) return tplhip_fail(TPLHIP_EINVAL, "bad reference line argument");
    // RefLine(data, len, step_size), utils.hpp:184-214: copy, then centre on the bounding box
    env->ref_line.resize((size_t)n);
    if (n > 0) std::memcpy(env->ref_line.data(), ref_line, sizeof(RefPoint) * (size_t)n);
    env->ref_step = step_size;
    double x_min = INFINITY, x_max = -INFINITY, y_min = INFINITY, y_max = -INFINITY;
    for (const RefPoint& p : env->ref_line) {
        x_min = std::min(x_min, p.x); x_max = std::max(x_max, p.x);
        y_min = std::min(y_min, p.y); y_max = std::max(y_max, p.y);
    }
    env->x_offset = x_min + (x_max - x_min) / 2.0;
    env->y_offset = y_min + (y_max - y_min) / 2.0;
    for (RefPoint& p : env->ref_line) { p.x -= env->x_offset; p.y -= env->y_offset; }
    return TPLHIP_OK;
}

int tplhip_dp_env_ref_line_offset(tplhip_dp_env* env, double* x_offset, double* y_offset)
{
    if (!env || !x_offset || !y_offset) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    *x_offset = env->x_offset;
    *y_offset = env->y_offset;
    return TPLHIP_OK;
}

int tplhip_dp_env_insert_geometry(tplhip_dp_env* env, const double* geom, int n, int stationary)
{
    if (!env || (!geom && n > 0) || n < 0) return tplhip_fail(TPLHIP_EINVAL, "bad geometry argument");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    if (env->ref_line.empty())       // env.cu:355-358
        return tplhip_fail(TPLHIP_ESTATE, "cannot insert dyn object points with ref_line of 0 steps");
    const TplDpEnvParams& ep = env->params;
    env->vertices.reserve(env->vertices.size() + (size_t)n);
    for (int i = 0; i < n; ++i) {
        const double* v = geom + 3 * (size_t)i;
        // env.cu:362-366
        const int t_idx = (v[2] < (double)ep.dt_start)
                              ? 0
                              : (int)((double)roundf((float)((v[2] - (double)ep.dt_start) / (double)ep.dt)) + 1.0);
        uint32_t flags = (t_idx >= 0 && t_idx < 32) ? (1u << t_idx) : 0u;
        if (stationary) flags |= 1u << 31;
        EnvVertex e;
        e.x = (float)(v[0] - env->x_offset);
        e.y = (float)(v[1] - env->y_offset);
        e.flags = flags;
        env->vertices.push_back(e);
    }
    return TPLHIP_OK;
}

// updateDistMapLon (env.cu:95-129) on the Frenet occupancy currently in d_occ
static int launch_dist_map_lon(tplhip_dp_env* env, hipStream_t st)
{
    const TplDpEnvParams& ep = env->params;
    // LDS per workgroup: 12 bytes per (station, slice); at most 60 KB
    const int tc_cap = std::max(1, std::min(DML_TC, (int)(60 * 1024 / (12 * (size_t)ep.s_steps))));
    if (12 * (size_t)ep.s_steps * tc_cap > 64 * 1024)
        return tplhip_fail(TPLHIP_EINVAL, "s_steps = %d is too large for the distance-map kernel", ep.s_steps);
    hipLaunchKernelGGL(k_dist_map_lon, dim3(ep.l_steps, (ep.t_steps + tc_cap - 1) / tc_cap), dim3(256),
                       sizeof(float) * 3 * (size_t)ep.s_steps * tc_cap, st, env->d_dist_lon, env->d_occ, ep, tc_cap);
    return TPLHIP_OK;
}

int tplhip_dp_env_update_dist_maps(tplhip_dp_env* env)
{
    if (!env) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    if (!env->have_occ) return tplhip_fail(TPLHIP_ESTATE, "no occupancy map: call set_occ_map or update first");
    if (env->params.l_steps > 1024) return tplhip_fail(TPLHIP_EINVAL, "l_steps = %d exceeds one workgroup", env->params.l_steps);
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    // planners work on their own non-blocking streams and may still read the tables of this environment (plan_resident):
    // nothing here is ordered against them implicitly
    HIP_TRY(hipDeviceSynchronize(), "waiting for the device failed");
    hipStream_t st = nullptr;
    int rc;
    if ((rc = launch_dist_map_lon(env, st))) return rc;
    HIP_TRY(hipGetLastError(), "distance-map launch failed");
    HIP_TRY(hipStreamSynchronize(st), "updating the distance maps failed");
    return TPLHIP_OK;
}

int tplhip_dp_env_update(tplhip_dp_env* env)
{
    if (!env) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    const TplDpEnvParams& ep = env->params;
    // env.cu:383-389
    if ((float)env->ref_line.size() * env->ref_step < ep.s_max) {
        return tplhip_fail(TPLHIP_ESTATE, "refline length = %s < than environment s_max = %s",
                           std::to_string((float)env->ref_line.size() * env->ref_step).c_str(),
                           std::to_string(ep.s_max).c_str());
    }
    if (ep.l_steps > 1024) return tplhip_fail(TPLHIP_EINVAL, "l_steps = %d exceeds one workgroup", ep.l_steps);
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    // planners work on their own non-blocking streams and may still read the tables of this environment (plan_resident):
    // nothing here is ordered against them implicitly
    HIP_TRY(hipDeviceSynchronize(), "waiting for the device failed");

    // occupancy_renderer.cu:277-279
    const float size = ep.s_max + ep.l_max * 2.0f;
    const int W = (int)std::ceil(size * PPM);
    if (W < 1 || W > 32768) return tplhip_fail(TPLHIP_EINVAL, "occupancy image of %d pixels per side", W);
    const float scale = (float)(1.0 / ((double)size * 0.5));          // glUniform1f(scale, 1.0/(size*0.5))

    int rc;
    if ((rc = ensure_capacity(&env->d_cart, &env->cart_cap, (size_t)W * W, "occupancy image allocation failed"))) return rc;
    const size_t n_vtx = env->vertices.size() - env->vertices.size() % 3;   // glDrawArrays(GL_TRIANGLES) drops a partial triangle
    if ((rc = ensure_capacity(&env->d_vtx, &env->vtx_cap, n_vtx, "vertex buffer allocation failed"))) return rc;
    const size_t n_ref = env->ref_line.size();
    if ((rc = ensure_capacity(&env->d_semantic, &env->sem_cap, n_ref, "reference line allocation failed"))) return rc;

    // per-station frames and the float semantic channel of the reference line texture
    std::vector<StationFrame> frames((size_t)ep.s_steps);
    for (int is = 0; is < ep.s_steps; ++is) {
        const float s = ep.s_min + (float)is * ep._s_step_size;
        frames[(size_t)is] = station_frame(env->ref_line, env->ref_step, s);
    }
    std::vector<float> semantic(n_ref);
    for (size_t i = 0; i < n_ref; ++i) semantic[i] = (float)env->ref_line[i].semantic;

    hipStream_t st = nullptr;
    if (env->cart_staged) {                                 // the last image may still be on its way out of d_cart
        HIP_TRY(hipEventSynchronize(env->ev_cart), "copying the occupancy image failed");
        env->cart_staged = false;
    }
    HIP_TRY(hipMemcpyAsync(env->d_frames, frames.data(), sizeof(StationFrame) * frames.size(), hipMemcpyHostToDevice, st),
            "copying station frames failed");
    HIP_TRY(hipMemcpyAsync(env->d_semantic, semantic.data(), sizeof(float) * n_ref, hipMemcpyHostToDevice, st),
            "copying ref line to gpu failed");
    HIP_TRY(hipMemsetAsync(env->d_cart, 0, sizeof(uint32_t) * (size_t)W * W, st), "occupancy image clear failed");
    if (n_vtx > 0) {
        HIP_TRY(hipMemcpyAsync(env->d_vtx, env->vertices.data(), sizeof(EnvVertex) * n_vtx, hipMemcpyHostToDevice, st),
                "copying occupancy vertex data failed");
        hipLaunchKernelGGL(k_raster_or, dim3((unsigned)(n_vtx / 3)), dim3(64), 0, st,
                           env->d_cart, W, env->d_vtx, (int)(n_vtx / 3), scale);
    }
    // image -> page-locked host copy, beside the Frenet stages
    env->cart_staged = false;
    {
        const size_t px = (size_t)W * W;
        bool ok = true;
        if (px > env->h_cart_cap) {
            if (env->h_cart) (void)hipHostFree(env->h_cart);
            env->h_cart = nullptr; env->h_cart_cap = 0;
            ok = hipHostMalloc(&env->h_cart, sizeof(uint32_t) * px, hipHostMallocDefault) == hipSuccess;
            if (ok) env->h_cart_cap = px; else (void)hipGetLastError();
        }
        ok = ok && (env->copy_stream || hipStreamCreateWithFlags(&env->copy_stream, hipStreamNonBlocking) == hipSuccess);
        ok = ok && (env->ev_raster || hipEventCreateWithFlags(&env->ev_raster, hipEventDisableTiming) == hipSuccess);
        ok = ok && (env->ev_cart || hipEventCreateWithFlags(&env->ev_cart, hipEventDisableTiming) == hipSuccess);
        if (ok) {
            HIP_TRY(hipEventRecord(env->ev_raster, st), "event record failed");
            HIP_TRY(hipStreamWaitEvent(env->copy_stream, env->ev_raster, 0), "stream wait failed");
            HIP_TRY(hipMemcpyAsync(env->h_cart, env->d_cart, sizeof(uint32_t) * px, hipMemcpyDeviceToHost, env->copy_stream),
                    "copying the occupancy image failed");
            HIP_TRY(hipEventRecord(env->ev_cart, env->copy_stream), "event record failed");
            env->cart_staged = true;
        }
    }
    hipLaunchKernelGGL(k_occ_to_frenet, dim3(ep.s_steps), dim3(256), 0, st,
                       env->d_occ, env->d_cart, W, env->d_frames, ep);
    hipLaunchKernelGGL(k_close_intersections, dim3((ep.t_steps * ep.s_steps + 255) / 256), dim3(256), 0, st,
                       env->d_occ, env->d_semantic, (int)n_ref, env->ref_step, ep);
    if ((rc = launch_dist_map_lon(env, st))) return rc;
    HIP_TRY(hipGetLastError(), "occupancy pipeline launch failed");
    HIP_TRY(hipStreamSynchronize(st), "updating the environment failed");
    env->vertices.clear();            // occupancy_renderer.cu:311
    env->cart_px = W;
    env->rendered = true;
    env->have_occ = true;
    return TPLHIP_OK;
}

int tplhip_dp_env_cart_size(tplhip_dp_env* env, int* size_pixels)
{
    if (!env || !size_pixels) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    *size_pixels = env->rendered ? env->cart_px : 1;      // FrameBuffer(1, 1) before the first render
    return TPLHIP_OK;
}

int tplhip_dp_env_get_occ_map_cartesian(tplhip_dp_env* env, uint32_t* out)
{
    if (!env || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!env->rendered) { out[0] = 0u; return TPLHIP_OK; }
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    if (env->cart_staged) {                                 // update() already has it on the way to the host
        HIP_TRY(hipEventSynchronize(env->ev_cart), "copying the occupancy image failed");
        std::memcpy(out, env->h_cart, sizeof(uint32_t) * (size_t)env->cart_px * env->cart_px);
        return TPLHIP_OK;
    }
    return tplhip_copy_to_host(out, env->d_cart, sizeof(uint32_t) * (size_t)env->cart_px * env->cart_px, nullptr,
                               "copying the occupancy image failed");
}

int tplhip_dp_env_get_dist_map_lon(tplhip_dp_env* env, float* out)
{
    if (!env || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    const size_t n = (size_t)env->params.l_steps * env->params.s_steps * env->params.t_steps * 2;
    return tplhip_copy_to_host(out, env->d_dist_lon, n * sizeof(float), nullptr,
                               "copying longitudinal distance map from gpu to cpu failed");
}

namespace {
// [l][s][t][c] -> [t][l][s][c]: the time slices of a Frenet table below each other
__global__ void k_time_slices(float* __restrict__ out, const float* __restrict__ in, int L, int S, int T, int C)
{
    const size_t n = (size_t)L * S * T * C;
    for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < n; o += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(o % C);
        size_t r = o / C;
        const int s = (int)(r % S); r /= S;
        const int l = (int)(r % L);
        const int t = (int)(r / L);
        out[o] = in[(((size_t)l * S + s) * T + t) * C + c];
    }
}

int time_slices(tplhip_dp_env* env, const float* table, int channels, float* out, const char* what)
{
    if (!env || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    const TplDpEnvParams& ep = env->params;
    const size_t n = (size_t)ep.l_steps * ep.s_steps * ep.t_steps * channels;
    int rc;
    if ((rc = ensure_capacity(&env->d_slices, &env->slices_cap, n, "time slice buffer allocation failed"))) return rc;
    hipLaunchKernelGGL(k_time_slices, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, nullptr,
                       env->d_slices, table, ep.l_steps, ep.s_steps, ep.t_steps, channels);
    HIP_TRY(hipGetLastError(), what);
    return tplhip_copy_to_host(out, env->d_slices, n * sizeof(float), nullptr, what);
}
} // namespace

int tplhip_dp_env_get_occ_map_slices(tplhip_dp_env* env, float* out)
{
    return time_slices(env, env ? env->d_occ : nullptr, 1, out, "copying occupancy map failed");
}

int tplhip_dp_env_get_dist_map_lon_slices(tplhip_dp_env* env, float* out)
{
    return time_slices(env, env ? env->d_dist_lon : nullptr, 2, out, "copying longitudinal distance map from gpu to cpu failed");
}

int tplhip_dp_env_get_dist(tplhip_dp_env* env, float t, float s, float l, float* out)
{
    if (!env || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    float* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(float)), "allocation failed");
    hipLaunchKernelGGL(k_get_dist, dim3(1), dim3(1), 0, nullptr, d, env->d_occ, t, s, l, env->params);
    hipError_t e = hipMemcpy(out, d, sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(e, "reading the distance field failed");
    return TPLHIP_OK;
}

} // extern "C"
