// poly_lat.hip -- SURVEY 8(f-2): the lateral quintic-lattice planner that produces the path the DP
// (s,v,a) planner runs on, for gfx950.  Replaces library/src/dyn_prog/poly_lat_planner.cu behind
// the reference's PolyLatPlanner / PolyLatTraj surface (bindings_dp_poly_lat.cu).
//
// Re-design, not a translation:
//  * the reference walks its l_dst x s_dst x s node lattice with five kernels and a device sync
//    (local costs; path times with ONE THREAD per path; collisions; path sums with one thread
//    per path; selection) through a float4 3-D surface.  Here one launch does all per-path
//    work: a workgroup owns one (l_dst, s_dst) path, its lanes the stations; node values stay
//    in LDS between the phases, the two order-dependent float accumulations (travel time,
//    cost sum) are the only serial steps, and a second one-thread launch selects the path;
//  * the reference-line frame of a station (interpRefLine incl. sin/cos of the heading) is
//    computed once per station on the host in the reference's float/double mix instead of once
//    per node in two kernels;
//  * the final trajectory is built on the host from the selected indices like the reference's.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "dp_env_internal.h"
#include "dp_lon_math.h"

using namespace tplhip;

// ---- PolyQuintic, poly_interp.cu:96-168, 384-416 (host and device) -----------------------------

namespace {

__host__ __device__ inline void quintic_init(TplPolyQuintic* q, double x0, double y0, double dy0, double ddy0,
                                             double x1, double y1, double dy1, double ddy1)
{
    q->x0 = x0; q->x1 = x1; q->d = x1 - x0;
    const double d = q->d;
    // calc_poly5_coeffs(y0/d, dy0, ddy0*d, y1/d, dy1, ddy1*d)
    const double a0 = y0 / d, a0d = dy0, a0dd = ddy0 * d, a1 = y1 / d, a1d = dy1, a1dd = ddy1 * d;
    const double h0 = (1.0 / 2.0) * a0dd;
    const double h1 = (3.0 / 2.0) * a0dd;
    const double h2 = (1.0 / 2.0) * a1dd;
    q->c[0] = a0;
    q->c[1] = a0d;
    q->c[2] = h0;
    q->c[3] = -h1 + h2 - 10 * a0 - 6 * a0d + 10 * a1 - 4 * a1d;
    q->c[4] = h1 + 15 * a0 + 8 * a0d - 15 * a1 + 7 * a1d - a1dd;
    q->c[5] = -h0 + h2 - 6 * a0 - 3 * a0d + 6 * a1 - 3 * a1d;
}

__host__ __device__ inline double quintic_eval(const TplPolyQuintic& q, int order, double x)
{
    const double t = (x - q.x0) / q.d;
    const double* c = q.c;
    switch (order) {
    case 0: return (t * (t * (t * (t * (t * c[5] + c[4]) + c[3]) + c[2]) + c[1]) + c[0]) * q.d;
    case 1: return t * (t * (t * (5 * t * c[5] + 4 * c[4]) + 3 * c[3]) + 2 * c[2]) + c[1];
    case 2: return (t * (t * (20 * t * c[5] + 12 * c[4]) + 6 * c[3]) + 2 * c[2]) / q.d;
    default: return (t * (60 * t * c[5] + 24 * c[4]) + 6 * c[3]) / (q.d * q.d);
    }
}

__host__ __device__ inline float sqf(float x) { return x * x; }

// interpDistField, env.cu:225-234: filtered lookup in the Frenet occupancy, rules R2/R3
__device__ __forceinline__ float dist_field(const float* __restrict__ occ, const TplDpEnvParams& ep,
                                            float t, float s, float l)
{
    const float t_idx = env_t_idx(ep, t);
    const float sc = (s - ep.s_min) / (ep.s_max - ep.s_min) * (float)(ep.s_steps - 1);
    const float lc = (l - ep.l_min) / (ep.l_max - ep.l_min) * (float)(ep.l_steps - 1);
    const Tap tt = make_tap(t_idx + 0.5f, ep.t_steps);
    const Tap ts = make_tap(sc + 0.5f, ep.s_steps);
    const Tap tl = make_tap(lc + 0.5f, ep.l_steps);
    auto at = [&](int il, int is, int it) { return occ[((size_t)il * ep.s_steps + is) * ep.t_steps + it]; };
    const float b00 = lerp1(at(tl.i0, ts.i0, tt.i0), at(tl.i1, ts.i0, tt.i0), tl.w);
    const float b10 = lerp1(at(tl.i0, ts.i1, tt.i0), at(tl.i1, ts.i1, tt.i0), tl.w);
    const float b01 = lerp1(at(tl.i0, ts.i0, tt.i1), at(tl.i1, ts.i0, tt.i1), tl.w);
    const float b11 = lerp1(at(tl.i0, ts.i1, tt.i1), at(tl.i1, ts.i1, tt.i1), tl.w);
    return lerp1(lerp1(b00, b10, ts.w), lerp1(b01, b11, ts.w), tt.w);
}

// ------------------------------------------------------------------------
// One workgroup per (l_dst, s_dst) path, lanes over stations.  Phases as the reference's
// kernels: evalLocalPathCosts :10-103, calcPathTimes :105-147, checkCollisions :149-185,
// evalPathCosts :187-222.  nodes: float4[l_dst][s_dst][s]; path_res: float2[l_dst][s_dst] =
// (min collision station, path cost).
// LDS: float l[S] | v[S] | w[S] | z[S] | x[S] | y[S]
// ------------------------------------------------------------------------
__global__ void k_poly_lat_paths(float4* __restrict__ nodes, float2* __restrict__ path_res,
                                 const StationFrame* __restrict__ frames, const float* __restrict__ occ,
                                 TplPolyLatTrajPoint start, TplPolyLatParams p, TplDpEnvParams ep)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = p.s_steps;
    float* sh_l = reinterpret_cast<float*>(smem);
    float* sh_v = sh_l + S;
    float* sh_w = sh_v + S;
    float* sh_z = sh_w + S;
    float* sh_x = sh_z + S;
    float* sh_y = sh_x + S;
    double* sh_q = reinterpret_cast<double*>(smem + sizeof(float) * 6 * (size_t)((S + 1) & ~1));   // 8-byte aligned

    const int idx_l_dst = blockIdx.x;
    const int idx_s_dst = blockIdx.y;
    const float l_dst = p.l_dst_min + p._l_dst_step_size * (float)idx_l_dst;
    const float s_dst = p.s_dst_min + p._s_dst_step_size * (float)idx_s_dst;

    TplPolyQuintic poly;
    quintic_init(&poly, (double)start.s, (double)start.l, (double)start.dl, (double)start.ddl,
                 (double)(start.s + s_dst), (double)l_dst, 0.0, 0.0);

    // ---- local path costs, one station per lane
    for (int idx_s = threadIdx.x; idx_s < S; idx_s += blockDim.x) {
        const float s = start.s + p.s_min + p._s_step_size * (float)idx_s;
        float pl, pdl, pddl, pdddl;
        if (s >= start.s + s_dst) {
            pl = l_dst; pdl = 0.0f; pddl = 0.0f; pdddl = 0.0f;
        } else {
            pl = (float)quintic_eval(poly, 0, (double)s);
            pdl = (float)quintic_eval(poly, 1, (double)s);
            pddl = (float)quintic_eval(poly, 2, (double)s);
            pdddl = (float)quintic_eval(poly, 3, (double)s);
        }
        const StationFrame rp = frames[idx_s];
        const float heading_frenet = atanf(pdl);
        // p.k = (ddl / (dl*dl + 1.0) + rp.k) * cos(heading_frenet) / (1.0 - l * rp.k)  -> float
        const float pk = (float)(((double)pddl / ((double)(pdl * pdl) + 1.0) + rp.k) * (double)cosf(heading_frenet)
                                 / (1.0 - (double)pl * rp.k));
        const float k_abs_path = fmaxf(fabsf(pk), (float)fabs(rp.k));
        float pv = (float)rp.v_max;
        if (k_abs_path > 1e-6f) pv = fminf(pv, sqrtf(p.a_lat_abs_max / k_abs_path));

        float constr = 0.0f;
        {
            // max(0.0, sq(min(rp.v_max, start.v)) * fabsf(p.k) - a_lat_abs_max): all in double
            const double vm = fmin(rp.v_max, (double)start.v);
            constr = (float)((double)constr + fmax(0.0, (vm * vm) * (double)fabsf(pk) - (double)p.a_lat_abs_max));
        }
        if (s <= start.s + s_dst) {
            const double half_w = (double)(p.width_veh * 0.5f) * sqrt(2.0);
            constr = (float)((double)constr + fmax(0.0, (double)(fabsf(pk) - p.k_abs_max)));
            constr = (float)((double)constr + fmax(0.0, (double)pl - (rp.d_left - half_w)));
            constr = (float)((double)constr + fmax(0.0, (-rp.d_right + half_w) - (double)pl));
        }
        float cost = 0.0f;
        cost += p.w_dl * sqf(pdl);
        cost += p.w_ddl * sqf(pddl);
        cost += p.w_dddl * sqf(pdddl);
        if (fabsf(pk) > fabsf((float)rp.k)) cost += p.w_k * sqf(pk);
        cost = (float)((double)cost + 10e6 * (double)constr);

        sh_l[idx_s] = pl;
        sh_v[idx_s] = pv;
        sh_w[idx_s] = cost;
        // calcPathTimes: Cartesian position of the node
        sh_x[idx_s] = (float)(rp.x - rp.sin_h * (double)pl);
        sh_y[idx_s] = (float)(rp.y + rp.cos_h * (double)pl);
    }
    __syncthreads();
    // ---- travel time along the path: per-station increments in parallel, the running float sum
    //      in station order by one lane
    for (int idx_s = threadIdx.x; idx_s < S; idx_s += blockDim.x) {
        float inc = 0.0f;
        if (idx_s > 0) {
            const float dx = sh_x[idx_s] - sh_x[idx_s - 1], dy = sh_y[idx_s] - sh_y[idx_s - 1];
            inc = sqrtf(dx * dx + dy * dy);
        }
        // the quotient of the running sum's step, in parallel: the chain below is then one
        // double addition per station instead of an IEEE double division
        sh_q[idx_s] = (double)inc / fmax(1.0, (double)sh_v[idx_s]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        sh_z[0] = 0.0f;
        int idx_s = 1;
        for (; idx_s + 8 <= S; idx_s += 8) {
            double q8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) q8[k] = sh_q[idx_s + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                t = (float)((double)t + q8[k]);
                sh_z[idx_s + k] = t;
            }
        }
        for (; idx_s < S; ++idx_s) {
            t = (float)((double)t + sh_q[idx_s]);
            sh_z[idx_s] = t;
        }
    }
    __syncthreads();
    // ---- collisions
    for (int idx_s = threadIdx.x; idx_s < S; idx_s += blockDim.x) {
        const float s = start.s + p.s_min + p._s_step_size * (float)idx_s;
        const float l = sh_l[idx_s], t = sh_z[idx_s];
        float dist_sem = 0.0f;
        for (float t_sweep = -1.0f; (double)t_sweep < 1.1; t_sweep = (float)((double)t_sweep + 1.0)) {
            dist_sem = fmaxf(dist_sem, dist_field(occ, ep, t + t_sweep, s, l));
            dist_sem = fmaxf(dist_sem, dist_field(occ, ep, t + t_sweep, s, (float)((double)l + 0.25)));
            dist_sem = fmaxf(dist_sem, dist_field(occ, ep, t + t_sweep, s, (float)((double)l - 0.25)));
        }
        const bool collision = dist_sem > 0.0f;
        const float z = (collision && (double)t < 8.0 && s > p.length_veh) ? s : 10000.0f;
        sh_z[idx_s] = z;
        nodes[((size_t)idx_l_dst * p.s_dst_steps + idx_s_dst) * S + idx_s] = make_float4(l, sh_v[idx_s], z, sh_w[idx_s]);
    }
    __syncthreads();
    // ---- path cost: float sum in station order
    if (threadIdx.x == 0) {
        float traj_cost = 0.0f;
        float collision_dist = 1000.0f;
        for (int idx_s = 0; idx_s < S; ++idx_s) {
            traj_cost += sh_w[idx_s];
            collision_dist = fminf(collision_dist, sh_z[idx_s]);
        }
        if ((double)l_dst < -0.1) traj_cost += p.w_right;
        const float diff_l = l_dst - p.l_trg;
        traj_cost += p.w_l * diff_l * diff_l;
        traj_cost += p.w_len * fabsf(s_dst);
        path_res[idx_l_dst * p.s_dst_steps + idx_s_dst] = make_float2(collision_dist, traj_cost);
        // "first node holds costs for entire path"
        float4* n0 = nodes + ((size_t)idx_l_dst * p.s_dst_steps + idx_s_dst) * S;
        n0->z = collision_dist;
        n0->w = traj_cost;
    }
}

// selectPath, poly_lat_planner.cu:224-266
__global__ void k_poly_lat_select(int* __restrict__ best, const float2* __restrict__ path_res_g, TplPolyLatParams p)
{
    // the two passes are order dependent (running maximum, first minimum) and stay on one lane,
    // but they read the path results from LDS, where one coalesced pass of the wavefront put them
    // (2 x 273 dependent reads of global memory were 39 us of a 70 us update)
    extern __shared__ __attribute__((aligned(8))) float2 path_res[];
    const int n_paths = p.l_dst_steps * p.s_dst_steps;
    const int n_pad = (n_paths + 7) & ~7;            // padding entries: infinite cost, ignored by both passes
    for (int i = threadIdx.x; i < n_pad; i += blockDim.x) path_res[i] = i < n_paths ? path_res_g[i] : make_float2(0.0f, INFINITY);
    __syncthreads();
    if (threadIdx.x != 0) return;
    int min_idx_l_dst = p.l_dst_steps / 2 + 1;
    int min_idx_s_dst = p.s_dst_steps - 1;
    // (the reference reads this node through clamp addressing; l_dst_steps/2 + 1 can be one past
    // the end for l_dst_steps < 3)
    const int cl = min(min_idx_l_dst, p.l_dst_steps - 1);
    float max_collision_dist = path_res[cl * p.s_dst_steps + min_idx_s_dst].x;
    float min_cost = INFINITY;
    // (both passes walk the paths in the reference's order, l_dst outer, s_dst inner; eight entries
    // are read from LDS before the order-dependent updates run over them, and the updates are
    // selects, not branches: on one lane a taken branch costs more than the arithmetic it skips.
    // Padding entries carry an infinite cost, which both passes ignore.)
    for (int i0 = 0; i0 < n_paths; i0 += 8) {
        float2 n8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) n8[k] = path_res[i0 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 n = n8[k];
            const bool upd = !((double)n.y >= 1e6) && n.x > max_collision_dist + p.length_veh;
            max_collision_dist = upd ? n.x : max_collision_dist;
        }
    }
    int min_i = -1;
    for (int i0 = 0; i0 < n_paths; i0 += 8) {
        float2 n8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) n8[k] = path_res[i0 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 n = n8[k];
            const bool ok = !(fabs((double)(n.x - max_collision_dist)) > 1.0) && n.y < min_cost;
            min_cost = ok ? n.y : min_cost;
            max_collision_dist = ok ? n.x : max_collision_dist;
            min_i = ok ? i0 + k : min_i;
        }
    }
    if (min_i >= 0) { min_idx_l_dst = min_i / p.s_dst_steps; min_idx_s_dst = min_i - min_idx_l_dst * p.s_dst_steps; }
    best[0] = min_idx_l_dst;
    best[1] = min_idx_s_dst;
}

// shortAngleDist<float> / normalizeAngle<float>, utils.hpp:21-53: the template keeps float
// variables and lets the double literals promote the individual operations
float normalize_angle_f(float x)
{
    x = (float)std::fmod((double)x, 2 * M_PI);
    x = (float)std::fmod((double)x + 2 * M_PI, 2 * M_PI);
    if ((double)x > M_PI) x = (float)((double)x - 2 * M_PI);
    return x;
}

float short_angle_dist_f(float x, float y)
{
    x = normalize_angle_f(x);
    y = normalize_angle_f(y);
    float a0 = y - x;
    const float a1 = (float)((double)(y - x) + 2 * M_PI);
    const float a2 = (float)((double)(y - x) - 2 * M_PI);
    if (std::fabs(a1) < std::fabs(a0)) a0 = a1;
    if (std::fabs(a2) < std::fabs(a0)) a0 = a2;
    return a0;
}

void poly_lat_update_step_sizes(TplPolyLatParams* p)     // poly_lat_planner.cuh:45-56
{
    if (p->s_steps > 1) p->_s_step_size = (p->s_max - p->s_min) / (float)(p->s_steps - 1);
    if (p->l_dst_steps > 1) p->_l_dst_step_size = (p->l_dst_max - p->l_dst_min) / (float)(p->l_dst_steps - 1);
    if (p->s_dst_steps > 1) p->_s_dst_step_size = (p->s_dst_max - p->s_dst_min) / (float)(p->s_dst_steps - 1);
}

} // namespace

struct tplhip_poly_lat {
    int device = 0;
    TplPolyLatParams params;
    bool have_params = false;
    float4* d_nodes = nullptr;          // [l_dst][s_dst][s]
    float2* d_path_res = nullptr;       // [l_dst][s_dst]
    StationFrame* d_frames = nullptr;   // [s]
    int* d_best = nullptr;              // [2]
    bool have_nodes = false;
};

extern "C" {

void tplhip_poly_lat_params_default(TplPolyLatParams* p)
{
    // poly_lat_planner.cuh:13-43
    p->l_min = -5.0f; p->l_max = 5.0f; p->s_min = 0.0f; p->s_max = 200.0f;
    p->s_steps = 201;
    p->l_dst_min = -5.0f; p->l_dst_max = 5.0f; p->s_dst_min = 10.0f; p->s_dst_max = 70.0f;
    p->l_dst_steps = 21; p->s_dst_steps = 13;
    p->l_trg = 0.0f;
    p->w_l = 1.0f; p->w_k = 0.1f; p->w_dl = 0.0f; p->w_ddl = 0.0f; p->w_dddl = 1.0f; p->w_right = 0.0f;
    p->w_len = 0.0001f;
    p->k_abs_max = 1.0f; p->a_lat_abs_max = 2.5f;
    p->width_veh = 2.0f; p->length_veh = 2.0f;
    p->_s_step_size = 0.0f; p->_l_dst_step_size = 0.0f; p->_s_dst_step_size = 0.0f;
}

void tplhip_poly_quintic_init(TplPolyQuintic* q, double x0, double y0, double dy0, double ddy0,
                              double x1, double y1, double dy1, double ddy1)
{
    quintic_init(q, x0, y0, dy0, ddy0, x1, y1, dy1, ddy1);
}

double tplhip_poly_quintic_eval(const TplPolyQuintic* q, int order, double x)
{
    return quintic_eval(*q, order, x);
}

int tplhip_poly_lat_create(int device, tplhip_poly_lat** out)
{
    if (!out) return tplhip_fail(TPLHIP_EINVAL, "null out pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return tplhip_fail(TPLHIP_EDEVICE, "no usable HIP device");
    if (device < 0) device = tplhip_default_device(n);
    if (device >= n) return tplhip_fail(TPLHIP_EDEVICE, "no usable HIP device (requested %d)", device);
    tplhip_poly_lat* h = new (std::nothrow) tplhip_poly_lat();
    if (!h) return tplhip_fail(TPLHIP_EDEVICE, "out of host memory");
    h->device = device;
    tplhip_poly_lat_params_default(&h->params);
    *out = h;
    return TPLHIP_OK;
}

int tplhip_poly_lat_destroy(tplhip_poly_lat* h)
{
    if (!h) return TPLHIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_nodes); (void)hipFree(h->d_path_res); (void)hipFree(h->d_frames); (void)hipFree(h->d_best);
    delete h;
    return TPLHIP_OK;
}

int tplhip_poly_lat_reinit(tplhip_poly_lat* h, TplPolyLatParams* ps)
{
    if (!h || !ps) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (ps->s_steps < 2 || ps->s_steps > 1024 || ps->l_dst_steps < 1 || ps->s_dst_steps < 1)
        return tplhip_fail(TPLHIP_EINVAL, "lateral planner needs 2 <= s_steps <= 1024, l_dst_steps >= 1, s_dst_steps >= 1");
    if ((size_t)ps->l_dst_steps * ps->s_dst_steps > 8000)
        return tplhip_fail(TPLHIP_EINVAL, "lateral planner supports at most 8000 lattice paths (%d x %d requested)",
                           ps->l_dst_steps, ps->s_dst_steps);
    poly_lat_update_step_sizes(ps);
    if (h->have_params && std::memcmp(ps, &h->params, sizeof(TplPolyLatParams)) == 0) return TPLHIP_OK;
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    h->params = *ps;
    h->have_params = true;
    (void)hipFree(h->d_nodes); (void)hipFree(h->d_path_res); (void)hipFree(h->d_frames); (void)hipFree(h->d_best);
    h->d_nodes = nullptr; h->d_path_res = nullptr; h->d_frames = nullptr; h->d_best = nullptr;
    h->have_nodes = false;
    const size_t n_paths = (size_t)ps->l_dst_steps * ps->s_dst_steps;
    HIP_TRY(hipMalloc(&h->d_nodes, sizeof(float4) * n_paths * ps->s_steps), "path node allocation failed");
    HIP_TRY(hipMalloc(&h->d_path_res, sizeof(float2) * n_paths), "path result allocation failed");
    HIP_TRY(hipMalloc(&h->d_frames, sizeof(StationFrame) * (size_t)ps->s_steps), "station frame allocation failed");
    HIP_TRY(hipMalloc(&h->d_best, sizeof(int) * 2), "selection buffer allocation failed");
    return TPLHIP_OK;
}

int tplhip_poly_lat_update(tplhip_poly_lat* h, const TplPolyLatTrajPoint* start, tplhip_dp_env* env,
                           TplPolyLatTrajPoint* out_points, TplPolyQuintic* out_poly,
                           int* best_idx_l_dst, int* best_idx_s_dst)
{
    if (!h || !start || !env || !out_points || !out_poly) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!h->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the lateral planner");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    if (env->ref_line.empty()) return tplhip_fail(TPLHIP_ESTATE, "the environment holds no reference line");
    if (env->device != h->device) return tplhip_fail(TPLHIP_EINVAL, "planner and environment live on different devices");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const TplPolyLatParams& p = h->params;
    const int S = p.s_steps;

    std::vector<StationFrame> frames((size_t)S);
    for (int is = 0; is < S; ++is) {
        const float s = start->s + p.s_min + p._s_step_size * (float)is;
        frames[(size_t)is] = station_frame(env->ref_line, env->ref_step, s);
    }
    hipStream_t st = nullptr;
    HIP_TRY(hipMemcpyAsync(h->d_frames, frames.data(), sizeof(StationFrame) * frames.size(), hipMemcpyHostToDevice, st),
            "copying station frames failed");
    const int threads = ((S + 63) / 64) * 64;
    hipLaunchKernelGGL(k_poly_lat_paths, dim3(p.l_dst_steps, p.s_dst_steps), dim3(std::min(threads, 1024)),
                       sizeof(float) * 6 * (size_t)((S + 1) & ~1) + sizeof(double) * (size_t)S, st,
                       h->d_nodes, h->d_path_res, h->d_frames, env->d_occ, *start, p, env->params);
    hipLaunchKernelGGL(k_poly_lat_select, dim3(1), dim3(64), sizeof(float2) * (((size_t)p.l_dst_steps * p.s_dst_steps + 7) & ~(size_t)7), st,
                       h->d_best, h->d_path_res, p);
    HIP_TRY(hipGetLastError(), "evaluating local path costs failed");
    int best[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(best, h->d_best, sizeof(best), hipMemcpyDeviceToHost, st), "copying best_idx_l_dst failed");
    HIP_TRY(hipStreamSynchronize(st), "selecting path failed");
    h->have_nodes = true;
    if (best_idx_l_dst) *best_idx_l_dst = best[0];
    if (best_idx_s_dst) *best_idx_s_dst = best[1];

    // final path, poly_lat_planner.cu:441-483
    const float l_dst = p.l_dst_min + p._l_dst_step_size * (float)best[0];
    const float s_dst = p.s_dst_min + p._s_dst_step_size * (float)best[1];
    quintic_init(out_poly, (double)start->s, (double)start->l, (double)start->dl, (double)start->ddl,
                 (double)(start->s + s_dst), (double)l_dst, 0.0, 0.0);
    for (int idx_s = 0; idx_s < S; ++idx_s) {
        TplPolyLatTrajPoint pt;
        std::memset(&pt, 0, sizeof pt);
        pt.s = start->s + p.s_min + p._s_step_size * (float)idx_s;
        if (pt.s >= start->s + s_dst) {
            pt.l = l_dst; pt.dl = 0.0f; pt.ddl = 0.0f; pt.dddl = 0.0f;
        } else {
            pt.l = (float)quintic_eval(*out_poly, 0, (double)pt.s);
            pt.dl = (float)quintic_eval(*out_poly, 1, (double)pt.s);
            pt.ddl = (float)quintic_eval(*out_poly, 2, (double)pt.s);
            pt.dddl = (float)quintic_eval(*out_poly, 3, (double)pt.s);
        }
        const StationFrame rp = station_frame(env->ref_line, env->ref_step, pt.s);
        const float heading_frenet = std::atan(pt.dl);
        pt.x = env->x_offset + rp.x - rp.sin_h * (double)pt.l;
        pt.y = env->y_offset + rp.y + rp.cos_h * (double)pt.l;
        pt.heading = (float)((double)heading_frenet + rp.heading);
        pt.v = (float)rp.v_max;
        out_points[idx_s] = pt;
    }
    return tplhip_poly_lat_traj_update(out_points, S);
}

int tplhip_poly_lat_copy_nodes(tplhip_poly_lat* h, float* out)
{
    if (!h || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!h->have_nodes) return tplhip_fail(TPLHIP_ESTATE, "no lattice evaluated yet");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const size_t n = (size_t)h->params.l_dst_steps * h->params.s_dst_steps * h->params.s_steps;
    return tplhip_copy_to_host(out, h->d_nodes, sizeof(float4) * n, nullptr, "copying path nodes failed");
}

int tplhip_poly_lat_traj_update(TplPolyLatTrajPoint* points, int n)
{
    // PolyLatTraj::updateTimeDistCurv, poly_lat_planner.cu:300-320
    if (!points || n < 0) return tplhip_fail(TPLHIP_EINVAL, "bad trajectory argument");
    double dist = 0.0, t = 0.0;
    for (int i = 1; i < n + 1; ++i) {
        TplPolyLatTrajPoint& p = points[i - 1];
        p.distance = dist;
        p.t = (float)t;
        if (i == n) { p.k = 0.0f; break; }          // the reference reads points[n] here (undefined)
        const TplPolyLatTrajPoint& pn = points[i];
        const double dx = pn.x - p.x, dy = pn.y - p.y;
        const double d = std::sqrt(dx * dx + dy * dy);
        p.k = (float)((double)short_angle_dist_f(p.heading, pn.heading) / d);
        dist += d;
        t += d / (double)p.v;
    }
    return TPLHIP_OK;
}

int tplhip_poly_lat_traj_lerp(const TplPolyLatTrajPoint* pts, int n, double distance, TplPolyLatTrajPoint* out)
{
    // PolyLatTraj::lerp, poly_lat_planner.cu:270-295 with InterpVars utils.hpp:59-86
    if (!pts || !out || n < 2) return tplhip_fail(TPLHIP_EINVAL, "trajectory needs at least two points");
    int i_prev = 0, i_next = 0;
    if (distance < pts[0].distance) {
        i_prev = 0; i_next = 1;
    } else if (distance >= pts[n - 1].distance) {
        i_prev = n - 2; i_next = n - 1;
    } else {
        for (int i = 1; i < n; ++i) {
            if (pts[i - 1].distance <= distance && distance < pts[i].distance) { i_prev = i - 1; i_next = i; break; }
        }
    }
    double a = (distance - pts[i_prev].distance) / (pts[i_next].distance - pts[i_prev].distance);
    a = std::max(0.0, std::min(1.0, a));
    const double ai = (double)(1.0f - a);           // "double ai = 1.0f - i.a"
    const TplPolyLatTrajPoint& p = pts[i_prev];
    const TplPolyLatTrajPoint& q = pts[i_next];
    TplPolyLatTrajPoint r;
    std::memset(&r, 0, sizeof r);
    r.t = (float)((double)p.t * ai + (double)q.t * a);
    r.l = (float)((double)p.l * ai + (double)q.l * a);
    r.dl = (float)((double)p.dl * ai + (double)q.dl * a);
    r.ddl = (float)((double)p.ddl * ai + (double)q.ddl * a);
    r.dddl = (float)((double)p.dddl * ai + (double)q.dddl * a);
    r.s = (float)((double)p.s * ai + (double)q.s * a);
    r.v = (float)((double)p.v * ai + (double)q.v * a);
    r.x = p.x * ai + q.x * a;
    r.y = p.y * ai + q.y * a;
    r.heading = (float)((double)p.heading + (double)short_angle_dist_f(p.heading, q.heading) * a);
    r.distance = p.distance * ai + q.distance * a;
    r.k = (float)((double)p.k * ai + (double)q.k * a);
    *out = r;
    return TPLHIP_OK;
}

// PolyLatTraj::lerp over an array of distances and PolyLatTraj::as_numpy (bindings_dp_poly_lat.cu:55-97): rows of
// 12 doubles (t, l, dl, ddl, dddl, s, v, x, y, heading, distance, k)
static void poly_lat_row(const TplPolyLatTrajPoint& p, double* r)
{
    r[0] = p.t; r[1] = p.l; r[2] = p.dl; r[3] = p.ddl; r[4] = p.dddl; r[5] = p.s; r[6] = p.v; r[7] = p.x; r[8] = p.y;
    r[9] = p.heading; r[10] = p.distance; r[11] = p.k;
}

int tplhip_poly_lat_traj_lerp_rows(const TplPolyLatTrajPoint* pts, int n, const double* distances, int m, double* out)
{
    if (!distances || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    for (int k = 0; k < m; ++k) {
        TplPolyLatTrajPoint r;
        const int rc = tplhip_poly_lat_traj_lerp(pts, n, distances[k], &r);
        if (rc) return rc;
        poly_lat_row(r, out + 12 * (size_t)k);
    }
    return TPLHIP_OK;
}

int tplhip_poly_lat_traj_rows(const TplPolyLatTrajPoint* pts, int n, double* out)
{
    if ((!pts && n > 0) || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    for (int k = 0; k < n; ++k) poly_lat_row(pts[k], out + 12 * (size_t)k);
    return TPLHIP_OK;
}

} // extern "C"
