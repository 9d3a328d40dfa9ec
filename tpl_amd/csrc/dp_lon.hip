// dp_lon.hip -- HP-1: DP (s,v,a) longitudinal planner, hand-written for gfx950.
//
// Replaces library/src/dyn_prog/lon_planner.cu (+ the updateDistMapPath slice
// of env.cu) of the reference behind the C-ABI of include/tplhip.h.  This is a
// re-design, not a translation:
//
//  * value tables are plain HBM arrays Node[t][a][v][s] (12-byte cells, s fastest); the
//    tex3D hardware filter of the reference is the software filter of
//    dp_lon_math.h (rules R2/R3);
//  * one backward stage = one launch of k_backward.  A workgroup owns whole
//    (v,a) rows.  Because the successor (v',a') of action k and its two blend
//    weights are the same for every station of a row, the row first blends the
//    four neighbouring next-stage rows into ONE row per action, staged in LDS
//    (coalesced 16 B/lane reads along s), and each cell then finishes with a
//    1-D lerp between two LDS entries.  Per-(row,action) constants come from a
//    descriptor table through scalar loads (SGPRs), per-(stage,station)
//    constants from a prelude table;
//  * stages are enqueued back to back on one HIP stream -- no host sync
//    between sweeps (the reference synchronises after every stage,
//    lon_planner.cu:358);
//  * the forward rollout is ONE launch: eight wavefronts per rollout, the 21 candidate jerks of a step across the lanes
//    of one and the 21 x 21 of the speculated next step across the other seven, lexicographic wave-level argmin.
// This file is the host driver; the kernels of the general path are in dp_lon_kernels.h, the plain-fp32 sweeps in
// dp_sweep.hip.
#include <hip/hip_runtime.h>

#include <chrono>
#include <algorithm>
#include <cmath>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "dp_lon_internal.h"
#include "tplhip_debug.h"
#include "tplhip_common.h"
#include "dp_env_internal.h"

using namespace tplhip;

#include "dp_lon_kernels.h"


// ==========================================================================
// host side
// ==========================================================================

struct tplhip_dp_lon {
    int device = 0;
    int n_cu = 0;                       // compute units of the device (0: not asked yet)
    hipStream_t stream = nullptr;
    TplDpLonParams params;
    bool have_params = false;

    Node* d_tables = nullptr;           // [t_steps][a][v][s], 12-byte cells: the table set of the most recent plan
    Node* d_table_set[2] = {nullptr, nullptr};   // [0] = the planner's tables; [1] = second set, allocated on demand, so
                                                 // that the rollout of plan k can run under the sweeps of plan k+1
    size_t table_cells = 0;
    hipStream_t fwd_stream = nullptr;   // forward rollouts + their downloads in the back-to-back (plan_resident) mode
    hipEvent_t ev_swept = nullptr, ev_fwd[2] = {nullptr, nullptr};
    bool fwd_pending[2] = {false, false};
    float* d_dist_fwd[2] = {nullptr, nullptr};   // the distance map a pending rollout reads (the environment's is rewritten)
    size_t dist_fwd_cap = 0;
    float* d_in_fwd[2] = {nullptr, nullptr};     // ... and its own copy of path + initial states: the next upload need not wait for it
    size_t in_fwd_cap = 0;                       // (floats)
    long plan_count = 0;
    bool overlap = false;               // set by plan_resident once the second table set exists
    ActDesc* d_acts = nullptr;          // [a*v][9]
    float* d_path = nullptr;            // [path_steps][7]
    float* d_pre_vmax = nullptr;        // [t_steps][s]
    float* d_pre_sdist = nullptr;       // [t_steps][s]
    double* d_pre_prog = nullptr;       // [s]
    TplLonState* d_inits = nullptr;
    TplLonState* d_trajs = nullptr;
    TplLonState* d_reeval = nullptr;     // [t_steps], reeval's own buffer (keeps the rollouts of the last plan intact)
    int cap_reeval = 0;
    int cap_rollouts = 0;
    int n_rollouts = 0;
    bool tables_valid = false;

    // launch geometry of k_backward
    int s_pad = 0, rows_per_block = 0, nblocks = 0, grid = 0, kchunk = 0;
    bool big = false;                    // the grid is several residency rounds deep: k_backward3v (else k_backward3s)
    // k_backward3v / 3s keep a RUNNING lexicographic argmin, which equals the reference's two-pass argmin
    // whenever no NaN can occur; that is guaranteed when every parameter and path value is finite
    // and below 1e6 in magnitude (see reinit).  Otherwise the plan runs k_backward<.., 1, ..> with
    // the geometry below.
    bool params_bounded = false, path_bounded = false, env_bounded = false;
    bool bounded_but_dt_start = false;   // params_bounded without its test of dt_start (the one field tpl moves every cycle)
    bool use_buf = false;                // buffer loads: one stage table addressable with 32-bit byte offsets
    int vp_s_pad = 0;                    // s_steps rounded up to whole wavefronts (k_backward3v / 3s)
    int tile_v = 8, tile_a = 8;          // traversal tiles, see k_backward
    int tile_v3 = 8;                     // strip height of k_backward3v (the skewed traversal has a working set of its own)
    int skew_q16 = 0;                    // k_backward3v: skew of the traversal along v + a dt / 2 (dp_sweep.hip), 16.16 pairs per a-row
    size_t lds_bytes = 0, lds1_bytes = 0;     // k_backward's LDS block at kchunk actions per round / at one

    // cheap division forms, only used when k_verify_div proved them exact (see k_backward)
    float rcp_path_step = 0.0f;
    float rcp_path_lo = 0.0f, rcp_range_hi = 0.0f, rcp_range_lo = 0.0f;   // two-float reciprocals (k_backward3v)
    int path_div3 = 0, range_div3 = 0;        // the plain-fp32 sweeps' forms (div_const in dp_sweep.hip): 4, 3, 5 proven, 0 = IEEE division
    float path_aux = 0.0f, range_aux = 0.0f;  // second operand of the chosen form: the reciprocal's low part (3) or the divisor (5, 0)
    int verified_path3 = 0, verified_range3 = 0;   // bit 0: form 3 proven, bit 1: form 4, bit 2: form 5
    bool weights_nonneg = false;              // every cost weight >= +0: (constraint, cost) compare as one integer key
    bool no_s3 = false;                       // TPLHIP_DEBUG=no_s3: small grids on the general sweep
    float verified_c[2] = {0.0f, 0.0f};      // divisors and range the proof covers
    float verified_xmax = 0.0f;
    int verified_s_steps = 0;                 // the coordinate check of the range divisor depends on s_steps - 1
    unsigned int* d_flag = nullptr;           // [8]: path {float, double, two-float, product}, range {the same}

    // two staged downloads in flight (pinned host buffers + completion events)
    TplLonState* h_stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    size_t stage_n[2] = {0, 0};          // states staged in each slot (rollouts x t_steps at staging time)
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};

    // page-locked staging of update()'s own transfers: path + initial states in (the copies are enqueued without a
    // host sync; `ev_up` guards the buffer against the next upload), trajectories out (one sync per plan)
    unsigned char* h_up = nullptr;
    size_t up_cap = 0;
    hipEvent_t ev_up = nullptr;
    bool up_inflight = false;
    unsigned int* h_flag = nullptr;      // page-locked word a single-rollout plan publishes its sequence number in (update())
    unsigned int flag_seq = 0;
    TplLonState* h_dl = nullptr;
    size_t dl_cap = 0;

    // scenario batches (tplhip_dp_lon_scenarios_*): n scenarios resident, `chunk` of them swept per launch
    int sc_n = 0, sc_chunk = 0, sc_paths = 0;
    TplDpEnvParams sc_ep;
    float* d_sc_occ = nullptr;           // [n][l][s][t]
    float* d_sc_path = nullptr;          // [sc_paths][path_steps][7]
    float* d_sc_dist = nullptr;          // [chunk][env t][env s]
    Node* d_sc_tables = nullptr;         // [chunk][t_steps][cells]
    float* d_sc_pre_vmax = nullptr;      // [chunk][t_steps][s]
    float* d_sc_pre_sdist = nullptr;
    TplLonState* d_sc_inits = nullptr;   // [n]
    TplLonState* d_sc_trajs = nullptr;   // [n][t_steps]
    unsigned char* d_sc_best = nullptr;  // TplDpPlanRecord + TplLonState[t_steps]
    bool sc_bounded = false;
    int sc_last_first = 0, sc_last_count = 0;
    bool v3_ok = false;                  // the two-rows-per-lane geometry of k_backward3v fits this grid
    int v3_grid = 0;
    V3Map* d_v3_map = nullptr;           // v3_map_block and action 0's first rows per workgroup of that grid (SweepArgs::tile_map)
    size_t v3_map_cap = 0;
    // chained sweep (SweepChain, dp_lon_internal.h): all stages of a plan behind the terminal one in ONE launch of k_backward3v
    int chain_mode = -1;                 // tplhip_dp_lon_set_chained_sweep: -1 the synchronous calls only (default), 0 never, 1 every plan
    bool chain_any = false;              // TPLHIP_DEBUG=chain_any: no minimum grid size for the chained launch
    bool chain_stall = false;            // TPLHIP_DEBUG=chain_stall: the next chained launch waits for flags nobody writes (tests)
    bool chain_fail_once = false;        // TPLHIP_DEBUG=chain_fail_once: the next status read reports a wait that gave up (tests)
    bool chain_ok = false;               // this grid and rule can be chained (set by reinit / apply_rule)
    bool chain_broken = false;           // a wait gave up once: never again on this handle
    int* d_chain_deps = nullptr;         // [v3_grid][CHAIN_DEPS]
    size_t chain_deps_cap = 0;
    unsigned* d_chain_flags = nullptr;   // [t_steps][pairs]
    size_t chain_flags_cap = 0;
    unsigned* d_chain_err = nullptr;     // [0] a wait gave up, [1] workgroups whose dependencies overflow CHAIN_DEPS
    unsigned* h_chain_err = nullptr;     // page-locked, device-mapped: set with d_chain_err[0], read by chain_check without a copy
    unsigned chain_epoch = 0;
    bool chain_unchecked = false;        // a chained launch went out since d_chain_err was last read
    unsigned long long chain_launches = 0, chain_failures = 0;
    bool no_v3 = false, no_fwd_overlap = false, no_fwd_spec = false, no_front = false;   // TPLHIP_DEBUG overrides, read by reinit
    bool front_fused = true;                 // what the last plan did (info)
    FilterRule rule = kDefaultRule;      // R1 - R3 of the arithmetic contract (tplhip_dp_lon_set_filter_rule / _set_contraction / _set_blend_order)
    bool rule_set = false;               // a rule other than kDefaultRule: the kernels that fold the default rule take it at run time
    bool rule_all_forms = true;          // the texture-filter defaults (R2 rint, R3 fma form, a -> v -> s) under EITHER setting of R1: the
                                         // plain-fp32 sweeps are built with every division form; any other rule takes (5, 5) or (0, 0)
    bool s3_first_early = false;         // k_backward3s requests its first round's rows ahead of its prologue (grids of >= 4 wavefronts per SIMD)
    int s3_kc = 0, s3_grid = 0;          // k_backward3s (grids that leave the chip mostly empty): actions per round (0 = not used), grid
    size_t s3_lds = 0;

    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double sweep_ms = 0.0;
    int64_t sweep_launches = 0;
};

namespace tplhip { thread_local bool t_sweep_probe = false; }

namespace {

void update_step_sizes(TplDpLonParams* p)      // lon_planner.cuh:45-50
{
    p->_s_step_size = (p->s_max - p->s_min) / (float)(p->s_steps - 1);
    p->_v_step_size = (p->v_max - p->v_min) / (float)(p->v_steps - 1);
    p->_a_step_size = (p->a_max - p->a_min) / (float)(p->a_steps - 1);
}

void env_update_step_sizes(TplDpEnvParams* p)  // env.cuh:39-43
{
    p->_l_step_size = (p->l_max - p->l_min) / (float)(p->l_steps - 1);
    p->_s_step_size = (p->s_max - p->s_min) / (float)(p->s_steps - 1);
}

int pick_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -1;
    if (device < 0) device = tplhip_default_device(n);
    if (device >= n) return -1;
    return device;
}

void free_planner_buffers(tplhip_dp_lon* h)
{
    (void)hipFree(h->d_table_set[0]); (void)hipFree(h->d_table_set[1]);
    h->d_table_set[0] = h->d_table_set[1] = nullptr; h->d_tables = nullptr;
    h->overlap = false; h->fwd_pending[0] = h->fwd_pending[1] = false;
    (void)hipFree(h->d_acts);      h->d_acts = nullptr;
    (void)hipFree(h->d_v3_map);    h->d_v3_map = nullptr; h->v3_map_cap = 0;
    (void)hipFree(h->d_chain_deps);  h->d_chain_deps = nullptr; h->chain_deps_cap = 0;
    (void)hipFree(h->d_chain_flags); h->d_chain_flags = nullptr; h->chain_flags_cap = 0;
    (void)hipFree(h->d_chain_err);   h->d_chain_err = nullptr;
    if (h->h_chain_err) (void)hipHostFree(h->h_chain_err);
    h->h_chain_err = nullptr;
    h->chain_ok = false;
    (void)hipFree(h->d_path);      h->d_path = nullptr;
    (void)hipFree(h->d_pre_vmax);  h->d_pre_vmax = nullptr;
    (void)hipFree(h->d_pre_sdist); h->d_pre_sdist = nullptr;
    (void)hipFree(h->d_pre_prog);  h->d_pre_prog = nullptr;
    h->tables_valid = false;
}

bool path_is_bounded(const float* path, int path_steps)
{
    bool ok = true;
    const size_t n_path = 7 * (size_t)path_steps;
    for (size_t i = 0; i < n_path; ++i) ok = ok && std::isfinite(path[i]) && std::fabs(path[i]) < 1e6f;
    return ok;
}

int ensure_rollouts(tplhip_dp_lon* h, int n)
{
    const int T = h->params.t_steps;
    if (n <= h->cap_rollouts && h->d_trajs) return TPLHIP_OK;
    (void)hipFree(h->d_inits); (void)hipFree(h->d_trajs);
    h->d_inits = nullptr; h->d_trajs = nullptr; h->cap_rollouts = 0;
    HIP_TRY(hipMalloc(&h->d_inits, sizeof(TplLonState) * (size_t)n), "rollout init allocation failed");
    HIP_TRY(hipMalloc(&h->d_trajs, sizeof(TplLonState) * (size_t)n * T), "trajectory allocation failed");
    h->cap_rollouts = n;
    return TPLHIP_OK;
}

int check_env(tplhip_dp_lon* h, tplhip_dp_env* env)
{
    if (!h || !env) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (!h->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the planner");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    if (!env->have_occ) return tplhip_fail(TPLHIP_ESTATE, "environment holds no occupancy map");
    if (env->device != h->device) return tplhip_fail(TPLHIP_EINVAL, "planner and environment live on different devices");
    return TPLHIP_OK;
}

template <bool TERMINAL, int KCHUNK, bool USE_BUF, bool ORDERED = false>
void launch_backward_b(tplhip_dp_lon* h, int idx_t, int nx)
{
    const size_t S = (size_t)h->params.s_steps;
    const size_t lds = TERMINAL ? 0 : h->lds_bytes;
    hipLaunchKernelGGL((k_backward<TERMINAL, KCHUNK, USE_BUF, ORDERED>), dim3(h->grid), dim3(h->s_pad * h->rows_per_block),
                       lds, h->stream,
                       h->d_tables + (size_t)nx * h->table_cells,
                       h->d_tables + (size_t)idx_t * h->table_cells,
                       h->d_acts, h->d_path,
                       h->d_pre_vmax + (size_t)idx_t * S, h->d_pre_sdist + (size_t)idx_t * S,
                       h->d_pre_prog, h->params, h->s_pad, h->rows_per_block, h->nblocks,
                       h->tile_v, h->tile_a, (size_t)0, (size_t)0, (size_t)0, h->rule);
}

template <bool TERMINAL, int KCHUNK>
void launch_backward(tplhip_dp_lon* h, int idx_t, int nx)
{
    // buffer loads need 32-bit byte offsets inside one stage table
    if (!TERMINAL && h->rule.order != 0)
        launch_backward_b<TERMINAL, KCHUNK, false, !TERMINAL>(h, idx_t, nx);      // eight taps per cell-action, plain loads
    else if (!TERMINAL && h->use_buf)
        launch_backward_b<TERMINAL, KCHUNK, true>(h, idx_t, nx);
    else
        launch_backward_b<TERMINAL, KCHUNK, false>(h, idx_t, nx);
}

// Workgroup size of k_forward: 512 = one step of speculation, for table sets the Infinity Cache holds (the speculated
// reads are 21 x as many); 64 = one step per round beyond that (see the kernel)
int fwd_threads(const tplhip_dp_lon* h)
{
    return (!h->no_fwd_spec && sizeof(Node) * h->table_cells * (size_t)h->params.t_steps <= ((size_t)256 << 20)) ? 512 : 64;
}

// The form of the two divisions by run-time constants in the plain-fp32 sweeps, from what k_verify_div proved.  The
// default filter rule -- under contracted arithmetic (R1 = 1, the default) and under one rounding per operation (R1 = 0) alike --
// is built with every pair of forms: the cheapest proven one per divisor.  The other rules are built with (5, 5) and (0, 0) only.
void pick_div_forms(tplhip_dp_lon* h)
{
    auto best = [](int mask) { return (mask & 2) ? 4 : ((mask & 1) ? 3 : ((mask & 4) ? 5 : 0)); };
    int pd = best(h->verified_path3), rd = best(h->verified_range3);
    // TPLHIP_DEBUG=div_form=N (tests, soaks): form N in {3, 4, 5} where it is proven, else IEEE division; N = 0: IEEE divisions
    // everywhere, under every rule; any other value is ignored (the planner picks by itself)
    int forced = 0;
    const bool forced_set = debug_opt("div_form", &forced) && (forced == 0 || forced == 3 || forced == 4 || forced == 5);
    if (forced_set) {
        auto only = [&](int mask) { return ((forced == 3 && (mask & 1)) || (forced == 4 && (mask & 2)) || (forced == 5 && (mask & 4))) ? forced : 0; };
        pd = only(h->verified_path3);
        rd = only(h->verified_range3);
    }
    if (!h->rule_all_forms) {
        const bool newton = (h->verified_path3 & 4) && (h->verified_range3 & 4) && (!forced_set || forced == 5);
        pd = rd = newton ? 5 : 0;
    }
    h->path_div3 = pd;
    h->range_div3 = rd;
    h->path_aux = pd == 3 ? h->rcp_path_lo : h->params.path_step_size;
    h->range_aux = rd == 3 ? h->rcp_range_lo : (h->params.s_max - h->params.s_min);
}

// k_backward3v (dp_sweep.hip): the two-rows-per-lane geometry with the plain-fp32 instruction stream.  Needs buffer
// loads and non-negative weights (integer-key argmin).  Every filter rule and every divisor runs on it: a division
// without a proven cheap form is an IEEE division (pick_div_forms); contracted arithmetic (rule R1 = 1,
// tplhip_dp_lon_set_contraction) is a template variant of the same kernels.
bool use_v3(const tplhip_dp_lon* h)
{
    if (h->rule.order != 0)
        return !h->no_v3 && h->use_buf && h->weights_nonneg && backward3o_built(h->rule)
               && backward3o_lds_bytes(h->vp_s_pad, h->params.path_steps) <= 64 * 1024;
    return !h->no_v3 && h->use_buf && h->weights_nonneg
           && backward3v_lds_bytes(h->vp_s_pad, h->params.path_steps) <= 64 * 1024;
}

bool launch_backward3(tplhip_dp_lon* h, int idx_t, int nx, int chain_stages = 0)
{
    const size_t S = (size_t)h->params.s_steps;
    SweepArgs a;
    a.next = h->d_tables + (size_t)nx * h->table_cells;
    a.out = h->d_tables + (size_t)idx_t * h->table_cells;
    a.acts = h->d_acts;
    a.path = h->d_path;
    a.pre_vmax = h->d_pre_vmax + (size_t)idx_t * S;
    a.pre_sdist = h->d_pre_sdist + (size_t)idx_t * S;
    a.pre_prog = h->d_pre_prog;
    a.p = h->params;
    a.s_pad = h->vp_s_pad;
    a.rcp_path_hi = h->rcp_path_step; a.rcp_path_lo = h->path_aux;
    a.rcp_range_hi = h->rcp_range_hi; a.rcp_range_lo = h->range_aux;
    a.tile_v = h->tile_v3; a.tile_a = h->tile_a; a.skew_q16 = h->skew_q16; a.tile_map = h->d_v3_map;
    a.stride_tables = 0; a.stride_pre = 0; a.stride_path = 0;
    dim3 grid(h->v3_grid, 1);
    if (chain_stages > 0) {
        if (!a.tile_map) return false;              // (a chained workgroup takes its first rows from the traversal entry, behind its wait)
        // stages idx_t, idx_t - 1, ... in ONE launch (SweepChain): grid.y = stage, the latest first
        a.chain.stages = chain_stages;
        a.chain.epoch = ++h->chain_epoch;
        a.chain.wait_epoch = a.chain.epoch;
        a.chain.flags = h->d_chain_flags;
        a.chain.first = idx_t;
        a.chain.pairs = h->params.a_steps * ((h->params.v_steps + 1) / 2);
        a.chain.deps = h->d_chain_deps;
        a.chain.err = h->d_chain_err;
        a.chain.err_host = h->h_chain_err;
        a.chain.table_cells = h->table_cells;
        a.chain.budget = 100000000;                 // 1 s of the 100 MHz counter: three orders of magnitude above any wait seen (a
                                                    // plan of this size takes 2 ms), short of every watchdog
        if (h->chain_stall) {
            // (tests: the flags every waiting workgroup looks for are never written -- each gives up after 20 us of the
            // counter or as soon as it sees that another has: the device side of chain_check's failure path)
            a.chain.wait_epoch = a.chain.epoch + 0x40000000u;
            a.chain.budget = 2000;
            h->chain_stall = false;
        }
        grid.y = (unsigned)chain_stages;
    }
    return launch_backward3v(h->rule, h->path_div3, h->range_div3, a, grid, backward3v_lds_bytes(h->vp_s_pad, h->params.path_steps), h->stream);
}

bool launch_backward3s_(tplhip_dp_lon* h, int idx_t, int nx)
{
    const size_t S = (size_t)h->params.s_steps;
    SweepArgs a;
    a.next = h->d_tables + (size_t)nx * h->table_cells;
    a.out = h->d_tables + (size_t)idx_t * h->table_cells;
    a.acts = h->d_acts;
    a.path = h->d_path;
    a.pre_vmax = h->d_pre_vmax + (size_t)idx_t * S;
    a.pre_sdist = h->d_pre_sdist + (size_t)idx_t * S;
    a.pre_prog = h->d_pre_prog;
    a.p = h->params;
    a.s_pad = h->vp_s_pad;
    a.rcp_path_hi = h->rcp_path_step; a.rcp_path_lo = h->path_aux;
    a.rcp_range_hi = h->rcp_range_hi; a.rcp_range_lo = h->range_aux;
    a.tile_v = h->tile_v; a.tile_a = h->tile_a; a.skew_q16 = h->skew_q16; a.tile_map = nullptr;
    a.stride_tables = 0; a.stride_pre = 0; a.stride_path = 0;
    return launch_backward3s(h->rule, h->path_div3, h->range_div3, h->s3_kc, h->s3_first_early, a, dim3(h->s3_grid, 1), h->s3_lds, h->stream);
}

// Which sweep a plan of this handle takes: 1 = k_backward3v, 2 = k_backward3s, 0 = the general k_backward.  Grids that
// fill the chip take the two-rows-per-lane form; grids that do not take one cell per lane where a row fits a
// 256-lane workgroup, and the two-rows form otherwise (rows of 257 .. 1024 stations: 300 x 37 x 20 runs k_backward3v).
int sweep_kind(const tplhip_dp_lon* h)
{
    const bool running_argmin_ok = h->params_bounded && h->path_bounded && h->env_bounded && use_v3(h);
    if (!running_argmin_ok) return 0;
    if (h->rule.order != 0) return h->v3_ok ? 1 : 0;          // k_backward3o on every grid (no one-cell-per-lane form of it)
    if (h->big) return h->v3_ok ? 1 : 0;
    if (h->s3_kc) return 2;
    return (h->v3_ok && !h->no_s3) ? 1 : 0;
}

// does a plan of this handle sweep its stages in one chained launch?
// (synchronous: the call returns with the plan -- one plan at a time on this handle by construction, where the chained launch
// wins 4-10 %; a caller that keeps several planners busy through plan_resident loses 4 % with it and gets it only on request)
bool chain_plan(const tplhip_dp_lon* h, bool synchronous)
{
    return h->chain_ok && !h->chain_broken && (h->chain_mode == 1 || (h->chain_mode < 0 && synchronous)) && sweep_kind(h) == 1
           && h->rule_all_forms && h->d_chain_deps && h->d_chain_flags && h->d_chain_err && h->h_chain_err;
}

void launch_stage(tplhip_dp_lon* h, int i, int nx, bool terminal)
{
    const int kind = sweep_kind(h);
    if (terminal) {
        launch_backward<true, 1>(h, i, nx);                       // rule R4 (no division in it)
    } else if (kind == 1 && launch_backward3(h, i, nx)) {
        // two rows per lane, plain-fp32 stream.  (A `false` means no instance is built for this (rule, division forms) pair:
        // pick_div_forms / backward3o_built keep the host's choices inside what dp_sweep.hip builds -- tests/test_capi.py::
        // test_every_selectable_rule_has_a_built_sweep -- and if they ever drift the stage is swept by the general kernel below
        // instead of being skipped.)
    } else if (kind == 2 && launch_backward3s_(h, i, nx)) {
        // the plain-fp32 stream, one cell per lane
    } else if (h->kchunk == 3) {
        launch_backward<false, 3>(h, i, nx);                 // the general sweep: two-pass argmin, any input
    } else {
        launch_backward<false, 1>(h, i, nx);
    }
}

// device part of DynProgLonPlanner::update (lon_planner.cu:341-388), all async on h->stream
// overlap: back-to-back mode (plan_resident).  Plans alternate between two table sets; the forward rollout runs on
// its own stream behind the last sweep, so the sweeps of the NEXT plan (main stream, other table set) do not wait
// for its ~50 dependent steps.  Without overlap everything is enqueued on the main stream and uses table set 0.
int wait_pending_forwards(tplhip_dp_lon* h)
{
    if (h->fwd_stream && (h->fwd_pending[0] || h->fwd_pending[1])) {
        HIP_TRY(hipStreamSynchronize(h->fwd_stream), "lon planner forward pass failed");
        h->fwd_pending[0] = h->fwd_pending[1] = false;
    }
    return TPLHIP_OK;
}

// A chained launch went out since the last look: did one of its waits give up (SweepChain)?  Called where a plan's results
// are handed out, behind the wait for them.  It cannot happen while workgroups start in the order of their ids; if it ever
// does, the plan's tables are not complete: the call fails, and the handle sweeps stage by stage from then on.
int chain_check(tplhip_dp_lon* h)
{
    if (!h->chain_unchecked) return TPLHIP_OK;
    h->chain_unchecked = false;
    // (the word is page-locked host memory the giving-up workgroup writes through its device mapping; every caller of this
    // function has just waited for the stream the launch ran on)
    unsigned err = __atomic_load_n(h->h_chain_err, __ATOMIC_ACQUIRE);
    if (h->chain_fail_once) { err = 1; h->chain_fail_once = false; }
#ifdef TPL_CHAIN_STATS
    {
        unsigned w[8];
        (void)hipMemcpy(w, h->d_chain_err, sizeof w, hipMemcpyDeviceToHost);
        unsigned long long ticks; std::memcpy(&ticks, w + 2, 8);
        std::fprintf(stderr, "[chain] launches %llu: waits %u workgroups, %u of them > 3 us, mean %.2f us\n", h->chain_launches, w[5], w[4], w[5] ? ticks / 100.0 / w[5] : 0.0);
        (void)hipMemset(h->d_chain_err + 2, 0, 4 * sizeof(unsigned));
    }
#endif
    if (err == 0) return TPLHIP_OK;
    h->chain_broken = true;
    h->chain_ok = false;
    h->chain_failures += 1;
    h->tables_valid = false;
    (void)hipMemset(h->d_chain_err, 0, sizeof(unsigned));
    *h->h_chain_err = 0;
    return tplhip_fail(TPLHIP_EDEVICE, "chained sweep: a workgroup waited a second for a row of the stage before it and gave up; this plan's "
                       "result is not valid -- plan again (the handle sweeps stage by stage from now on)");
}

int enqueue_plan(tplhip_dp_lon* h, tplhip_dp_env* env, bool overlap, TplLonState* mirror = nullptr,
                 unsigned int* done_flag = nullptr, unsigned int done_seq = 0)
{
    const TplDpLonParams& p = h->params;
    const int T = p.t_steps;
    const int par = overlap ? (int)(h->plan_count & 1) : 0;
    if (!overlap) {
        int rc = wait_pending_forwards(h);
        if (rc) return rc;
    } else if (h->fwd_pending[par]) {
        // the rollout of the plan before last still reads this table set
        HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_fwd[par], 0), "stream wait failed");
    }
    h->d_tables = h->d_table_set[par];
    h->plan_count += 1;
    {
        const TplDpEnvParams& e = env->params;
        const float vals[] = {e.s_min, e.s_max, e.l_min, e.l_max, e.dt, e.dt_start};
        bool ok = true;
        for (float v : vals) ok = ok && std::isfinite(v) && std::fabs(v) < 1e6f;
        h->env_bounded = ok;
    }

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        HIP_TRY(hipEventCreate(&e0), "event creation failed");
        HIP_TRY(hipEventCreate(&e1), "event creation failed");
    }
    // env.updateDistMapPath (lon_planner.cu:343), the stage prelude and the terminal stage (rule R4): one launch
    // (k_plan_front) where two distance slices fit the LDS of a block, else the three kernels it stands for
    const size_t front_lds = 2 * dist_slice_lds_bytes(env->params.s_steps);
    // (a plan of ONE stage launches no sweep at all -- the reference's loop, lon_planner.cu:347, does not run -- so nothing
    // may write a terminal table either: the fused front, whose terminal role always runs, is for T >= 2)
    const bool fused_front = front_lds <= 48 * 1024 && !h->no_front && T >= 2;      // (48 KB: the dynamic-LDS allowance of a launch without opt-in)
    h->front_fused = fused_front;
    // (the timed span of a plan starts HERE in both forms: distance map, prelude and the T - 1 stage launches -- so that
    // the microseconds per launch bench.py derives are comparable across no_front, large environments and rounds)
    if (h->timing) HIP_TRY(hipEventRecord(e0, h->stream), "event record failed");
    if (fused_front) {
        const int nrows = p.v_steps * p.a_steps;
        const int rpb = std::max(1, (nrows + 4 * h->n_cu - 1) / (4 * h->n_cu));
        const dim3 grid(env->params.t_steps + (T - 1) + (nrows + rpb - 1) / rpb);
        Node* term = h->d_tables + (size_t)(T - 1) * h->table_cells;
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, dim3(256), front_lds, h->stream, env->d_dist, env->d_occ, h->d_path,
                               h->d_pre_vmax, h->d_pre_sdist, h->d_pre_prog, term, p, env->params, rpb, h->rule);
        };
        if (h->rule_set) go(k_plan_front<true>); else go(k_plan_front<false>);
    } else {
        hipLaunchKernelGGL(k_dist_map_path, dim3(env->params.t_steps), dim3(256),
                           dist_slice_lds_bytes(env->params.s_steps), h->stream,
                           env->d_dist, env->d_occ, h->d_path, p.path_steps, p.path_step_size, env->params,
                           (size_t)0, (size_t)0, (size_t)0, h->rule);
        hipLaunchKernelGGL(k_stage_prelude, dim3(T), dim3(256), 0, h->stream,
                           h->d_pre_vmax, h->d_pre_sdist, h->d_pre_prog, h->d_path, env->d_dist, p, env->params,
                           (size_t)0, (size_t)0, (size_t)0, h->rule);
    }
    // backward pass, lon_planner.cu:347-361 -- no host sync between stages
    for (int i = T - 1; i > 0; --i) {
        const int nx = std::min(T - 1, i + 1);
        if (i == T - 1 && fused_front) continue;
        if (i == T - 2 && i >= 2 && chain_plan(h, mirror != nullptr) && launch_backward3(h, i, nx, i)) {
            // stages T-2 .. 1 in one launch (SweepChain)
            h->chain_unchecked = true;
            h->chain_launches += 1;
            break;
        }
        launch_stage(h, i, nx, i == T - 1);
    }
    if (h->timing) {
        HIP_TRY(hipEventRecord(e1, h->stream), "event record failed");
        h->events.emplace_back(e0, e1);
        h->sweep_launches += (T - 1);
    }
    // forward pass, lon_planner.cu:365-386 -- one launch
    {
        hipStream_t fs = h->stream;
        const float* dist = env->d_dist;
        const float* path_in = h->d_path;
        const TplLonState* inits_in = h->d_inits;
        if (overlap) {
            // everything the rollout reads besides its table set goes into buffers of the plan's parity, in ONE launch behind the
            // sweeps: the distance map (the environment's is rewritten by the next plan's front), the path and the initial
            // states (the next upload() overwrites them) -- a caller that replans with new inputs every cycle never waits for
            // a rollout on the host
            const unsigned dist_w = (unsigned)((size_t)env->params.t_steps * env->params.s_steps);
            const unsigned path_w = 7u * (unsigned)p.path_steps, init_w = (unsigned)(sizeof(TplLonState) / 4 * (size_t)h->n_rollouts);
            const unsigned blocks = std::min(64u, (dist_w + path_w + init_w + 255u) / 256u);
            hipLaunchKernelGGL(k_copy_fwd_inputs, dim3(blocks), dim3(256), 0, h->stream,
                               reinterpret_cast<const unsigned*>(env->d_dist), reinterpret_cast<unsigned*>(h->d_dist_fwd[par]), dist_w,
                               reinterpret_cast<const unsigned*>(h->d_path), reinterpret_cast<unsigned*>(h->d_in_fwd[par]), path_w,
                               reinterpret_cast<const unsigned*>(h->d_inits), reinterpret_cast<unsigned*>(h->d_in_fwd[par] + path_w), init_w);
            HIP_TRY(hipEventRecord(h->ev_swept, h->stream), "event record failed");
            HIP_TRY(hipStreamWaitEvent(h->fwd_stream, h->ev_swept, 0), "stream wait failed");
            fs = h->fwd_stream;
            dist = h->d_dist_fwd[par];
            path_in = h->d_in_fwd[par];
            inits_in = reinterpret_cast<const TplLonState*>(h->d_in_fwd[par] + path_w);
        }
        const size_t ring_b = sizeof(float) * 7 * FWD_RING;
        const size_t maps_b = sizeof(float) * ((size_t)p.path_steps * 7 + (size_t)env->params.t_steps * env->params.s_steps);
        {
            const bool in_lds = ring_b + maps_b <= 60 * 1024, rt = h->rule_set;
            auto go = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(h->n_rollouts), dim3(fwd_threads(h)), in_lds ? ring_b + maps_b : ring_b, fs,
                                   h->d_trajs, inits_in, h->d_tables, h->table_cells,
                                   path_in, dist, p, env->params, (size_t)0, (size_t)0, (size_t)0, h->rule, mirror, done_flag, done_seq);
            };
            if (fwd_threads(h) == 64) {
                if (in_lds) { if (rt) go(k_forward_seq<true, true>); else go(k_forward_seq<true, false>); }
                else { if (rt) go(k_forward_seq<false, true>); else go(k_forward_seq<false, false>); }
            } else if (in_lds) { if (rt) go(k_forward<true, true>); else go(k_forward<true, false>); }
            else { if (rt) go(k_forward<false, true>); else go(k_forward<false, false>); }
        }
        if (overlap) {
            HIP_TRY(hipEventRecord(h->ev_fwd[par], h->fwd_stream), "event record failed");
            h->fwd_pending[par] = true;
        }
    }
    HIP_TRY(hipGetLastError(), "lon planner launch failed");
    h->tables_valid = true;
    return TPLHIP_OK;
}

// second table set + rollout stream of the back-to-back mode; any failure just leaves the mode off
void try_enable_overlap(tplhip_dp_lon* h, tplhip_dp_env* env)
{
    const size_t dist_elems = (size_t)env->params.t_steps * env->params.s_steps;
    const size_t in_elems = 7 * (size_t)h->params.path_steps + sizeof(TplLonState) / 4 * (size_t)h->n_rollouts;
    if (h->no_fwd_overlap) { h->overlap = false; return; }
    if (h->overlap && h->dist_fwd_cap >= dist_elems && h->in_fwd_cap >= in_elems) return;
    h->overlap = false;
    const size_t bytes = sizeof(Node) * h->table_cells * (size_t)h->params.t_steps;
    if (!h->d_table_set[1]) {
        if (hipMalloc(&h->d_table_set[1], bytes) != hipSuccess) { (void)hipGetLastError(); h->d_table_set[1] = nullptr; return; }
        if (hipMemsetAsync(h->d_table_set[1], 0, bytes, h->stream) != hipSuccess) return;
    }
    if (!h->fwd_stream && hipStreamCreateWithFlags(&h->fwd_stream, hipStreamNonBlocking) != hipSuccess) { h->fwd_stream = nullptr; return; }
    if (!h->ev_swept && hipEventCreateWithFlags(&h->ev_swept, hipEventDisableTiming) != hipSuccess) return;
    for (int i = 0; i < 2; ++i)
        if (!h->ev_fwd[i] && hipEventCreateWithFlags(&h->ev_fwd[i], hipEventDisableTiming) != hipSuccess) return;
    if (h->dist_fwd_cap < dist_elems) {
        if (hipStreamSynchronize(h->fwd_stream) != hipSuccess) return;
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(h->d_dist_fwd[i]); h->d_dist_fwd[i] = nullptr;
            if (hipMalloc(&h->d_dist_fwd[i], sizeof(float) * dist_elems) != hipSuccess) { (void)hipGetLastError(); h->dist_fwd_cap = 0; return; }
        }
        h->dist_fwd_cap = dist_elems;
    }
    if (h->in_fwd_cap < in_elems) {
        if (hipStreamSynchronize(h->fwd_stream) != hipSuccess) return;
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(h->d_in_fwd[i]); h->d_in_fwd[i] = nullptr;
            if (hipMalloc(&h->d_in_fwd[i], sizeof(float) * in_elems) != hipSuccess) { (void)hipGetLastError(); h->in_fwd_cap = 0; return; }
        }
        h->in_fwd_cap = in_elems;
    }
    h->overlap = true;
}

} // namespace

void free_scenarios_fwd(tplhip_dp_lon* h);

extern "C" {

void tplhip_dp_lon_params_default(TplDpLonParams* p)
{
    // lon_planner.cuh:9-43
    p->s_min = 0.0f; p->s_max = 200.0f; p->v_min = 0.0f; p->v_max = 36.0f;
    p->a_min = -2.0f; p->a_max = 2.0f; p->j_min = -2.0f; p->j_max = 2.0f;
    p->t_steps = 10; p->s_steps = 201; p->v_steps = 37; p->a_steps = 7;
    p->dt_start = 1.0f; p->dt = 1.0f;
    p->_s_step_size = 0.0f; p->_v_step_size = 0.0f; p->_a_step_size = 0.0f;
    p->time_gap = 1.5f; p->gap_min = 1.0f;
    p->w_progress = 1.0f; p->w_a = 0.5f; p->w_j = 0.5f; p->w_snap = 0.5f; p->w_safety_dist = 10.0f;
    p->path_step_size = 0.5f; p->path_steps = 200;
    p->width_veh = 2.0f; p->length_veh = 6.0f;
}

void tplhip_dp_env_params_default(TplDpEnvParams* p)
{
    // env.cuh:13-37
    p->t_min = 0.0f; p->t_max = 10.0f; p->l_min = -5.0f; p->l_max = 5.0f;
    p->s_min = 0.0f; p->s_max = 200.0f; p->ds_max = 36.0f;
    p->dir_min = (float)-M_PI_2; p->dir_max = (float)M_PI_2;
    p->dt = 1.0f; p->dt_start = 1.0f;
    p->t_steps = 10; p->l_steps = 21; p->s_steps = 201; p->dir_steps = 21;
    p->scale_objects = 1.0f; p->dilation = 2.0f;
    p->_l_step_size = 0.0f; p->_s_step_size = 0.0f;
}

// ---- host-only helpers -----------------------------------------------------

void tplhip_lon_dynamics(const TplLonState* x0, float j, float dt, TplLonState* out)
{
    // lon_planner.cu:240-250 (host instantiation of the __interop__ function)
    TplLonState r;
    r.t = x0->t + dt;
    float sf = x0->s + x0->v * dt;
    double sd = (double)sf + ((0.5 * (double)x0->a) * (double)dt) * (double)dt;
    sd = sd + ((((1.0 / 6.0) * (double)j) * (double)dt) * (double)dt) * (double)dt;
    r.s = fmaxf(x0->s, (float)sd);
    float vf = x0->v + x0->a * dt;
    double vd = (double)vf + ((0.5 * (double)j) * (double)dt) * (double)dt;
    r.v = fmaxf(0.0f, (float)vd);
    r.a = x0->a + j * dt;
    r.j = j;
    r.cost = 0.0f;
    r.constr = 0.0f;
    *out = r;
}

int tplhip_lon_traj_state(const TplLonState* states, int n, float t, TplLonState* out)
{
    // LonTraj::state lon_planner.cu:252-260; bracket search InterpVars utils.hpp:59-78
    if (!states || !out || n < 2) return tplhip_fail(TPLHIP_EINVAL, "trajectory needs at least two states");
    const double v = (double)t;
    int i_prev = 0;
    if (v < (double)states[0].t) {
        i_prev = 0;
    } else if (v >= (double)states[n - 1].t) {
        i_prev = n - 2;
    } else {
        for (int i = 1; i < n; ++i) {
            if ((double)states[i - 1].t <= v && v < (double)states[i].t) { i_prev = i - 1; break; }
        }
    }
    const float t_rel = t - states[i_prev].t;
    tplhip_lon_dynamics(&states[i_prev], states[i_prev].j, t_rel, out);
    return TPLHIP_OK;
}

// LonTraj::state over an array of times (bindings_dp_lon.cu:93-108): rows (s, v, a, j)
int tplhip_lon_traj_state_rows(const TplLonState* states, int n, const double* ts, int m, double* out)
{
    if (!ts || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    for (int k = 0; k < m; ++k) {
        TplLonState r;
        const int rc = tplhip_lon_traj_state(states, n, (float)ts[k], &r);
        if (rc) return rc;
        out[4 * (size_t)k] = r.s; out[4 * (size_t)k + 1] = r.v; out[4 * (size_t)k + 2] = r.a; out[4 * (size_t)k + 3] = r.j;
    }
    return TPLHIP_OK;
}

// ---- environment -------------------------------------------------------------

int tplhip_dp_env_create(int device, tplhip_dp_env** out)
{
    if (!out) return tplhip_fail(TPLHIP_EINVAL, "null out pointer");
    const int dev = pick_device(device);
    if (dev < 0) return tplhip_fail(TPLHIP_EDEVICE, "no usable HIP device (requested %d)", device);
    tplhip_dp_env* e = new (std::nothrow) tplhip_dp_env();
    if (!e) return tplhip_fail(TPLHIP_EDEVICE, "out of host memory");
    e->device = dev;
    tplhip_dp_env_params_default(&e->params);
    *out = e;
    return TPLHIP_OK;
}

int tplhip_dp_env_destroy(tplhip_dp_env* env)
{
    if (!env) return TPLHIP_OK;
    (void)hipSetDevice(env->device);
    (void)hipFree(env->d_occ);
    (void)hipFree(env->d_dist);
    (void)hipFree(env->d_cart);
    if (env->copy_stream) { (void)hipStreamSynchronize(env->copy_stream); (void)hipStreamDestroy(env->copy_stream); }
    if (env->h_cart) (void)hipHostFree(env->h_cart);
    if (env->ev_raster) (void)hipEventDestroy(env->ev_raster);
    if (env->ev_cart) (void)hipEventDestroy(env->ev_cart);
    (void)hipFree(env->d_vtx);
    (void)hipFree(env->d_frames);
    (void)hipFree(env->d_semantic);
    (void)hipFree(env->d_dist_lon);
    (void)hipFree(env->d_slices);
    delete env;
    return TPLHIP_OK;
}

int tplhip_dp_env_reinit(tplhip_dp_env* env, TplDpEnvParams* ps)
{
    if (!env || !ps) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (ps->t_steps < 1 || ps->s_steps < 2 || ps->l_steps < 2)
        return tplhip_fail(TPLHIP_EINVAL, "environment needs t_steps >= 1, s_steps >= 2, l_steps >= 2");
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    env_update_step_sizes(ps);
    const bool full = !env->have_params || ps->t_steps != env->params.t_steps
                      || ps->l_steps != env->params.l_steps || ps->s_steps != env->params.s_steps;
    env->params = *ps;
    env->have_params = true;
    if (!full) return TPLHIP_OK;
    (void)hipFree(env->d_occ); (void)hipFree(env->d_dist); (void)hipFree(env->d_dist_lon); (void)hipFree(env->d_frames);
    env->d_occ = nullptr; env->d_dist = nullptr; env->d_dist_lon = nullptr; env->d_frames = nullptr;
    const size_t n_occ = (size_t)ps->l_steps * ps->s_steps * ps->t_steps;
    const size_t n_dist = (size_t)ps->t_steps * ps->s_steps;
    HIP_TRY(hipMalloc(&env->d_occ, n_occ * sizeof(float)), "occupancy map allocation failed");
    HIP_TRY(hipMalloc(&env->d_dist, n_dist * sizeof(float)), "distance map allocation failed");
    HIP_TRY(hipMalloc(&env->d_dist_lon, n_occ * 2 * sizeof(float)), "longitudinal distance map allocation failed");
    HIP_TRY(hipMalloc(&env->d_frames, sizeof(tplhip::StationFrame) * (size_t)ps->s_steps), "station frame allocation failed");
    HIP_TRY(hipMemset(env->d_dist_lon, 0, n_occ * 2 * sizeof(float)), "longitudinal distance map clear failed");
    // a freshly allocated cudaArray is read as all-free road until update() ran
    HIP_TRY(hipMemset(env->d_occ, 0, n_occ * sizeof(float)), "occupancy map clear failed");
    HIP_TRY(hipMemset(env->d_dist, 0, n_dist * sizeof(float)), "distance map clear failed");
    HIP_TRY(hipDeviceSynchronize(), "clearing the environment tables failed");   // planners read them from their own (non-blocking) streams
    env->have_occ = true;
    return TPLHIP_OK;
}

int tplhip_dp_env_set_occ_map(tplhip_dp_env* env, const float* occ_lst)
{
    if (!env || !occ_lst) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    // planners work on their own non-blocking streams and may still read the tables of this environment (plan_resident):
    // nothing here is ordered against them implicitly
    HIP_TRY(hipDeviceSynchronize(), "waiting for the device failed");
    const size_t n = (size_t)env->params.l_steps * env->params.s_steps * env->params.t_steps;
    HIP_TRY(hipMemcpy(env->d_occ, occ_lst, n * sizeof(float), hipMemcpyHostToDevice), "copying occupancy map failed");
    env->have_occ = true;
    return TPLHIP_OK;
}

int tplhip_dp_env_get_occ_map(tplhip_dp_env* env, float* occ_lst)
{
    if (!env || !occ_lst) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    const size_t n = (size_t)env->params.l_steps * env->params.s_steps * env->params.t_steps;
    return tplhip_copy_to_host(occ_lst, env->d_occ, n * sizeof(float), nullptr, "copying occupancy map failed");
}

int tplhip_dp_env_get_dist_map_path(tplhip_dp_env* env, float* dist_ts)
{
    if (!env || !dist_ts) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!env->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the environment");
    HIP_TRY(hipSetDevice(env->device), "hipSetDevice failed");
    // planners work on their own non-blocking streams and may still read the tables of this environment (plan_resident):
    // nothing here is ordered against them implicitly
    HIP_TRY(hipDeviceSynchronize(), "waiting for the device failed");
    const size_t n = (size_t)env->params.t_steps * env->params.s_steps;
    return tplhip_copy_to_host(dist_ts, env->d_dist, n * sizeof(float), nullptr, "copying distance map failed");
}

// ---- planner -----------------------------------------------------------------

int tplhip_dp_lon_create(int device, tplhip_dp_lon** out)
{
    if (!out) return tplhip_fail(TPLHIP_EINVAL, "null out pointer");
    const int dev = pick_device(device);
    if (dev < 0) return tplhip_fail(TPLHIP_EDEVICE, "no usable HIP device (requested %d)", device);
    HIP_TRY(hipSetDevice(dev), "hipSetDevice failed");
    tplhip_dp_lon* h = new (std::nothrow) tplhip_dp_lon();
    if (!h) return tplhip_fail(TPLHIP_EDEVICE, "out of host memory");
    h->device = dev;
    tplhip_dp_lon_params_default(&h->params);
    // non-blocking: nothing here is ordered against the null stream implicitly (the environment works on it and ends
    // every update with a host wait); implicit null-stream coupling costs sporadic 30 ms waits on this runtime
    hipError_t err = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (err != hipSuccess) {
        delete h;
        return tplhip_fail(TPLHIP_EDEVICE, "%s: stream creation failed", hipGetErrorString(err));
    }
    *out = h;
    return TPLHIP_OK;
}

int tplhip_dp_lon_destroy(tplhip_dp_lon* h)
{
    if (!h) return TPLHIP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->fwd_stream) { (void)hipStreamSynchronize(h->fwd_stream); (void)hipStreamDestroy(h->fwd_stream); }
    if (h->ev_swept) (void)hipEventDestroy(h->ev_swept);
    for (int i = 0; i < 2; ++i) { if (h->ev_fwd[i]) (void)hipEventDestroy(h->ev_fwd[i]); (void)hipFree(h->d_dist_fwd[i]); (void)hipFree(h->d_in_fwd[i]); }
    for (auto& ev : h->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    free_planner_buffers(h);
    free_scenarios_fwd(h);
    (void)hipFree(h->d_inits);
    (void)hipFree(h->d_trajs);
    (void)hipFree(h->d_reeval);
    (void)hipFree(h->d_flag);
    for (int i = 0; i < 2; ++i) {
        if (h->h_stage[i]) (void)hipHostFree(h->h_stage[i]);
        if (h->stage_ev[i]) (void)hipEventDestroy(h->stage_ev[i]);
    }
    if (h->h_up) (void)hipHostFree(h->h_up);
    if (h->h_dl) (void)hipHostFree(h->h_dl);
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    if (h->ev_up) (void)hipEventDestroy(h->ev_up);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return TPLHIP_OK;
}

namespace {
// The action table of h->params under h->rule and, behind it on the same stream, the traversal table that copies from it.
int build_action_tables(tplhip_dp_lon* h)
{
    const TplDpLonParams& p = h->params;
    const int nrows = p.v_steps * p.a_steps;
    hipLaunchKernelGGL(k_build_actions, dim3((nrows * N_BWD + 255) / 256), dim3(256), 0, h->stream, h->d_acts, p, h->rule);
    HIP_TRY(hipGetLastError(), "action table launch failed");
    h->chain_ok = false;
    if (h->d_v3_map) {
        // chained sweep: where asked for, for the default filter rule and blend order under either R1 (the instances dp_sweep.hip
        // builds the chained form for), rows of whole cache lines (SweepChain), four stages or more
        const int pairs = p.a_steps * ((p.v_steps + 1) / 2);
        // (not on stage tables beyond 96 MB: there the launch boundary is 1 % of a sweep and the write-through stores cost
        // 3 % -- 512x256x128: 325 -> 335 us, profiles/r06_dp_chained_sweep.txt)
        bool chain = h->chain_mode != 0 && !h->chain_broken && h->rule_all_forms && ((size_t)p.s_steps * sizeof(Node)) % 128 == 0
                     && p.t_steps >= 4 && h->table_cells * sizeof(Node) <= ((size_t)96 << 20)
                     // ... and not where a stage leaves half the chip empty: the workgroups of the stages behind it, resident and
                     // waiting, take the free slots, and every stage then runs packed on the compute units its turn found free
                     // instead of spread over all of them (256x40x52: 14.2 -> 19.1 us per stage sweep; from one residency round
                     // up the chained launch wins: 256x64x64 21.9 -> 20.3, 512x64x32 24.5 -> 22.0, 256x128x64 37.6 -> 34.3)
                     && ((size_t)h->v3_grid * (size_t)(h->vp_s_pad / 64) >= (size_t)h->n_cu * 32 || h->chain_any);
        if (chain) {
            const size_t nd = (size_t)h->v3_grid * CHAIN_DEPS, nf = (size_t)p.t_steps * pairs;
            if (nd > h->chain_deps_cap) {
                (void)hipFree(h->d_chain_deps); h->d_chain_deps = nullptr; h->chain_deps_cap = 0;
                if (hipMalloc(&h->d_chain_deps, sizeof(int) * nd) == hipSuccess) h->chain_deps_cap = nd; else { (void)hipGetLastError(); chain = false; }
            }
            if (chain && nf > h->chain_flags_cap) {
                (void)hipFree(h->d_chain_flags); h->d_chain_flags = nullptr; h->chain_flags_cap = 0;
                if (hipMalloc(&h->d_chain_flags, sizeof(unsigned) * nf) == hipSuccess) h->chain_flags_cap = nf; else { (void)hipGetLastError(); chain = false; }
                if (chain) HIP_TRY(hipMemsetAsync(h->d_chain_flags, 0, sizeof(unsigned) * nf, h->stream), "flag reset failed");
            }
            if (chain && !h->d_chain_err && hipMalloc(&h->d_chain_err, 8 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); h->d_chain_err = nullptr; chain = false; }
            if (chain && !h->h_chain_err) {
                if (hipHostMalloc(&h->h_chain_err, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) *h->h_chain_err = 0;
                else { (void)hipGetLastError(); h->h_chain_err = nullptr; chain = false; }
            }
            if (chain) HIP_TRY(hipMemsetAsync(h->d_chain_err, 0, 8 * sizeof(unsigned), h->stream), "flag reset failed");
        }
        hipLaunchKernelGGL(k_fill_v3_map, dim3((h->v3_grid + 255) / 256), dim3(256), 0, h->stream, h->d_v3_map, h->v3_grid, h->d_acts,
                           p.v_steps, p.a_steps, h->tile_v3, h->tile_a, h->skew_q16, chain ? h->d_chain_deps : nullptr,
                           chain ? h->d_chain_err + 1 : nullptr);
        HIP_TRY(hipGetLastError(), "traversal table launch failed");
        if (chain) {
            // (once per reinit of a chained planner: does every workgroup's dependency list fit?)
            unsigned over = 0;
            HIP_TRY(hipMemcpyAsync(&over, h->d_chain_err + 1, sizeof over, hipMemcpyDeviceToHost, h->stream), "dependency check failed");
            HIP_TRY(hipStreamSynchronize(h->stream), "dependency check failed");
            h->chain_ok = over == 0;
        }
    }
    return TPLHIP_OK;
}
} // namespace

int tplhip_dp_lon_reinit(tplhip_dp_lon* h, TplDpLonParams* ps)
{
    if (!h || !ps) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (ps->t_steps < 1 || ps->s_steps < 2 || ps->v_steps < 2 || ps->a_steps < 2 || ps->path_steps < 1)
        return tplhip_fail(TPLHIP_EINVAL, "planner needs t_steps >= 1, s/v/a_steps >= 2, path_steps >= 1");
    if (ps->s_steps > 1024)
        return tplhip_fail(TPLHIP_EINVAL, "s_steps = %d exceeds 1024 (one workgroup spans a row; the reference has the same limit)", ps->s_steps);
    if (ps->v_steps > 32767 || ps->a_steps > 32767)
        return tplhip_fail(TPLHIP_EINVAL, "v_steps / a_steps exceed 32767");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    update_step_sizes(ps);                                  // lon_planner.cu:286

    if (h->have_params && std::memcmp(ps, &h->params, sizeof(TplDpLonParams)) == 0) {
        return TPLHIP_OK;                                   // lon_planner.cu:288-290
    }
    // tpl moves dt_start every cycle and nothing else: the action table, the launch geometry and the division proofs do
    // not depend on it (k_build_actions integrates over params.dt; dt_start reaches the kernels as an argument)
    if (h->have_params && h->d_tables) {
        TplDpLonParams a = *ps, b = h->params;
        a.dt_start = 0.0f; b.dt_start = 0.0f;
        if (std::memcmp(&a, &b, sizeof a) == 0) {
            h->params.dt_start = ps->dt_start;
            h->params_bounded = h->bounded_but_dt_start && std::isfinite(ps->dt_start) && std::fabs(ps->dt_start) < 1e6f;
            h->tables_valid = false;
            return TPLHIP_OK;
        }
    }
    const bool resize = !h->have_params || !h->d_tables
        || ps->t_steps != h->params.t_steps || ps->s_steps != h->params.s_steps
        || ps->v_steps != h->params.v_steps || ps->a_steps != h->params.a_steps
        || ps->path_steps != h->params.path_steps;

    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    {
        int rc = wait_pending_forwards(h);
        if (rc) return rc;
    }
    h->params = *ps;
    h->have_params = true;
    h->tables_valid = false;
    const TplDpLonParams& p = h->params;
    const size_t S = (size_t)p.s_steps;
    const int nrows = p.v_steps * p.a_steps;

    if (resize) {
        free_planner_buffers(h);
        (void)hipFree(h->d_inits); (void)hipFree(h->d_trajs);
        h->d_inits = nullptr; h->d_trajs = nullptr; h->cap_rollouts = 0;
        h->n_rollouts = 0; h->path_bounded = false;          // nothing uploaded any more
        h->stage_busy[0] = h->stage_busy[1] = false;
        free_scenarios_fwd(h);                               // scenario tables are sized by the grid
        h->table_cells = S * p.v_steps * p.a_steps;
        HIP_TRY(hipMalloc(&h->d_table_set[0], sizeof(Node) * h->table_cells * (size_t)p.t_steps), "value table allocation failed");
        h->d_tables = h->d_table_set[0];
        HIP_TRY(hipMemsetAsync(h->d_tables, 0, sizeof(Node) * h->table_cells * (size_t)p.t_steps, h->stream), "value table clear failed");
        HIP_TRY(hipMalloc(&h->d_acts, sizeof(ActDesc) * (size_t)nrows * N_BWD), "action table allocation failed");
        HIP_TRY(hipMalloc(&h->d_path, sizeof(float) * 7 * (size_t)p.path_steps), "path array allocation failed");
        HIP_TRY(hipMalloc(&h->d_pre_vmax, sizeof(float) * S * p.t_steps), "prelude allocation failed");
        HIP_TRY(hipMalloc(&h->d_pre_sdist, sizeof(float) * S * p.t_steps), "prelude allocation failed");
        HIP_TRY(hipMalloc(&h->d_pre_prog, sizeof(double) * S), "prelude allocation failed");
    }

    // launch geometry
    h->s_pad = ((p.s_steps + 63) / 64) * 64;
    h->rows_per_block = std::max(1, 256 / h->s_pad);
    h->nblocks = (nrows + h->rows_per_block - 1) / h->rows_per_block;
    {
        // every XCD gets the blocks of its own v range (ranges differ by at most one v index)
        int max_rows = 0;
        for (int x = 0; x < 8; ++x)
            max_rows = std::max(max_rows, ((((x + 1) * p.v_steps) >> 3) - ((x * p.v_steps) >> 3)) * p.a_steps);
        h->grid = 8 * ((max_rows + h->rows_per_block - 1) / h->rows_per_block);
        // strip width: the widest (even, <= 8) whose working set -- 9 action windows x (tile_a + 1)
        // a-rows x (tile_v + 3) v-rows of the next-stage table -- stays inside an XCD's 4 MiB L2.
        // 512x256x128 (6 KB rows): 8 would need 5.3 MB and costs 4.5 % (386 vs 369 us per sweep).
        h->tile_a = 8;
        h->tile_v = 8;
        while (h->tile_v > 2 && (size_t)9 * (h->tile_a + 1) * (h->tile_v + 3) * p.s_steps * sizeof(Node) > (size_t)3584 * 1024)
            h->tile_v -= 2;
        int ov = 0;
        if (debug_opt("tile_v", &ov)) h->tile_v = std::max(1, ov);
        if (debug_opt("tile_a", &ov)) h->tile_a = std::max(1, ov);
    }
    const size_t threads = (size_t)h->s_pad * h->rows_per_block;
    const size_t path_lds = sizeof(float) * ((size_t)p.path_steps + 1);       // + the duplicate of the last sample
    const size_t lds_max = 64 * 1024;                       // keep >= 2 workgroups per CU
    // Actions blended into LDS per barrier.  With the L2-friendly traversal the sweep is bound by
    // VALU issue and by how well memory latency hides under it: 1 action per pass keeps the LDS
    // block at 4 KB and the kernel at ~50 VGPRs (8 waves per SIMD) and wins everywhere it was
    // measured (256x128x64: 55.0 us vs 56.2 (3) vs 61.8 (9); 512x256x128: 526 / 555 / 575 us).
    // A grid that fits the chip in one residency round (config 2, the reference default grid)
    // has nothing to overlap with and prefers fewer barriers: 3 per pass (10.7 vs 12.0 us).
    {
        if (h->n_cu <= 0) {                                 // asked once per planner: the query takes tens of microseconds
            h->n_cu = 256;
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && n > 0) h->n_cu = n;
        }
        const int n_cu = h->n_cu;
        const size_t waves = (size_t)h->nblocks * (threads / 64);
        // (measured per grid with tools/plan_time_grid.py: at 4096 waves the three-per-pass kernel wins by 10 %, at 8192
        // the two-rows-per-lane sweep by 7 %)
        h->kchunk = (waves <= (size_t)n_cu * 4 * 6) ? 3 : 1;
    }
    if (threads * 16 * 3 + path_lds > lds_max) h->kchunk = 1;
    h->big = h->kchunk == 1;                                // decided on the grid, before any override
    {
        int b = 0;                                          // soaks: the large-grid kernels on a small grid (big=1) or the reverse
        if (debug_opt("big", &b)) h->big = b != 0;
    }
    {
        int k = 0;                                          // tuning override: 3 or 1
        if (debug_opt("kchunk", &k) && (k == 3 || k == 1) && threads * 16 * (size_t)k + path_lds <= lds_max) h->kchunk = k;
    }
    h->lds_bytes = threads * 16 * (size_t)h->kchunk + path_lds;
    h->lds1_bytes = threads * 16 + path_lds;
    h->use_buf = h->table_cells * sizeof(Node) < 0xffffffffull && !debug_opt("no_bufload");
    h->no_v3 = debug_opt("no_v3");
    h->no_fwd_overlap = debug_opt("no_fwd_overlap");
    h->no_fwd_spec = debug_opt("no_fwd_spec");
    h->no_front = debug_opt("no_front");
    {
        int m = 0;
        if (debug_opt("chain", &m)) h->chain_mode = m != 0 ? 1 : 0;          // TPLHIP_DEBUG=chain=1 | chain=0
        h->chain_fail_once = debug_opt("chain_fail_once");
        h->chain_stall = debug_opt("chain_stall");
        h->chain_any = debug_opt("chain_any");               // tests / fuzzing: chain grids of any size (the bits do not depend on it)
    }
    // "bounded": every float parameter finite and < 1e6 in magnitude, at most 1e5 stages.  Then no
    // cost or constraint sum of the sweep can overflow (largest single term ~ w * (1e6)^2 = 1e18, at
    // most 1e5 of them), no infinity and hence no NaN can appear, and a running lexicographic argmin
    // is identical to the two-pass argmin of lon_planner.cu:153-170.
    {
        const float vals[] = {p.s_min, p.s_max, p.v_min, p.v_max, p.a_min, p.a_max, p.j_min, p.j_max, p.dt,
                              p.time_gap, p.gap_min, p.w_progress, p.w_a, p.w_j, p.w_snap, p.w_safety_dist,
                              p.path_step_size, p.width_veh, p.length_veh};
        bool ok = p.t_steps <= 100000;
        for (float v : vals) ok = ok && std::isfinite(v) && std::fabs(v) < 1e6f;
        // the plain-fp32 sweeps read the path's second tap as "the next slot": needs arclength >= 0
        ok = ok && p.s_min >= 0.0f && p.path_step_size > 0.0f;
        h->bounded_but_dt_start = ok;
        h->params_bounded = ok && std::isfinite(p.dt_start) && std::fabs(p.dt_start) < 1e6f;
        // k_backward3v orders (constraint, cost) by their bit patterns: needs both to be sums of terms >= +0
        bool nn = true;
        for (float w : {p.w_progress, p.w_a, p.w_j, p.w_snap, p.w_safety_dist}) nn = nn && !std::signbit(w);
        h->weights_nonneg = nn;
    }
    // k_backward3v: two rows per lane; row PAIRS are dealt to the XCDs, so it takes any v_steps
    {
        h->vp_s_pad = ((p.s_steps + 63) / 64) * 64;
        const int np = (p.v_steps + 1) / 2;
        int max_pairs = 0;
        for (int x = 0; x < 8; ++x) max_pairs = std::max(max_pairs, (((x + 1) * np) >> 3) - ((x * np) >> 3));
        h->v3_grid = 8 * max_pairs * p.a_steps;
        h->v3_ok = h->vp_s_pad <= 1024;
        // rows that read the same next-stage rows: v + a dt / 2 = const (see the kernel); in row PAIRS per a-row
        {
            // (used where the rectangular traversal already had to shrink its strips to stay inside L2 -- tables beyond ~100 MB;
            // on 256x128x64 both traversals move the same bytes and take the same time; TPLHIP_DEBUG=skew=0|1 forces either)
            int forced = -1;
            (void)debug_opt("skew", &forced);
            const bool on = forced < 0 ? h->tile_v < 8 : forced != 0;
            const double k = (p._v_step_size > 0.0f && std::isfinite(p.dt)) ? 0.5 * (double)p.dt * (double)p._a_step_size / (double)p._v_step_size / 2.0 : 0.0;
            h->skew_q16 = (on && k > 0.0 && k < 64.0) ? (int)std::lround(k * 65536.0) : 0;
            // Strip height of the skewed traversal: the workgroups resident on an XCD walk one strip down ALL a-rows
            // together and touch (strip + 2) v'-rows of every a'-row: the tallest even strip whose band stays inside the
            // XCD's 4 MiB L2 (512x256x128: 2 rows -- FETCH_SIZE 453 -> 275 MB per sweep, profiles/r04_dp_config4_skew.txt;
            // 256x128x64: 8, as before)
            h->tile_v3 = h->tile_v;
            int ov = 0;
            if (h->skew_q16 && !debug_opt("tile_v", &ov)) {
                h->tile_v3 = 8;
                while (h->tile_v3 > 2 && (size_t)p.a_steps * (h->tile_v3 + 2) * p.s_steps * sizeof(Node) > (size_t)3584 * 1024)
                    h->tile_v3 -= 2;
            }
            h->v3_ok = h->v3_ok && (h->tile_v3 % 2 == 0);
        }
        // the traversal as a table: one V3Map per workgroup (64 KB on 256x128x64), filled behind the action table (fill_v3_map)
        if (h->v3_ok && p.a_steps < 65536 && p.v_steps < 65536) {
            if ((size_t)h->v3_grid > h->v3_map_cap) {
                (void)hipFree(h->d_v3_map); h->d_v3_map = nullptr; h->v3_map_cap = 0;
                HIP_TRY(hipMalloc(&h->d_v3_map, sizeof(V3Map) * (size_t)h->v3_grid), "traversal table allocation failed");
                h->v3_map_cap = (size_t)h->v3_grid;
            }
        } else {
            (void)hipFree(h->d_v3_map); h->d_v3_map = nullptr; h->v3_map_cap = 0;
        }
    }
    // Grids that leave the chip mostly empty (the kchunk = 3 class above) with rows of at most 256 lanes: k_backward3s,
    // one cell per lane on the plain-fp32 instruction stream, three actions per barrier round (TPLHIP_DEBUG=no_s3: k_backward)
    {
        h->s3_kc = 0;
        const int want = 3;
        h->no_s3 = debug_opt("no_s3");
        if (!h->big && h->vp_s_pad <= 256 && !h->no_s3) {
            h->s3_lds = backward3s_lds_bytes(want, h->vp_s_pad, p.path_steps);
            int max_rows = 0;
            for (int x = 0; x < 8; ++x)
                max_rows = std::max(max_rows, ((((x + 1) * p.v_steps) >> 3) - ((x * p.v_steps) >> 3)) * p.a_steps);
            h->s3_grid = 8 * max_rows;
            if (h->s3_lds <= (size_t)64 * 1024) h->s3_kc = want;
            h->s3_first_early = (size_t)nrows * (h->vp_s_pad / 64) >= (size_t)h->n_cu * 4 * 4;
        }
    }
    if (h->lds_bytes > 160 * 1024)
        return tplhip_fail(TPLHIP_EINVAL, "path_steps = %d does not fit in LDS next to the row buffers", p.path_steps);

    int rc_acts = build_action_tables(h);
    if (rc_acts) return rc_acts;

    // Division by the two run-time constants of the sweep (path step, station range): the plain-fp32 sweeps use a
    // cheap form only if it is proven bit-identical to IEEE division for every float the sweep can feed it.
    // Largest dividend: s_max + the longest constant-jerk step, and |s_min|; the proof covers twice that.
    {
        const float s_range = p.s_max - p.s_min;
        const double dt = (double)p.dt;
        const double reach = std::fabs((double)p.v_max) * dt
                             + 0.5 * std::max(std::fabs((double)p.a_min), std::fabs((double)p.a_max)) * dt * dt
                             + std::max(std::fabs((double)p.j_min), std::fabs((double)p.j_max)) * dt * dt * dt / 6.0;
        const float xmax = (float)(2.0 * (std::max(std::fabs((double)p.s_min), std::fabs((double)p.s_max)) + (double)s_range + reach));
        h->rcp_path_step = 1.0f / p.path_step_size;
        h->rcp_path_lo = (float)(1.0 / (double)p.path_step_size - (double)h->rcp_path_step);
        h->rcp_range_hi = 1.0f / s_range;
        h->rcp_range_lo = (float)(1.0 / (double)s_range - (double)h->rcp_range_hi);
        const bool off = debug_opt("no_fastdiv");
        const bool cached = !off && h->verified_c[0] == p.path_step_size && h->verified_c[1] == s_range && h->verified_xmax >= xmax
                            && h->verified_s_steps == p.s_steps;
        if (!cached) {
            h->verified_path3 = h->verified_range3 = 0;
            h->verified_c[0] = h->verified_c[1] = 0.0f;
            if (!off && std::isfinite(xmax) && xmax > 0.0f && s_range > 0.0f && p.path_step_size > 0.0f) {
                if (!h->d_flag) HIP_TRY(hipMalloc(&h->d_flag, 6 * sizeof(unsigned int)), "flag allocation failed");
                HIP_TRY(hipMemsetAsync(h->d_flag, 0, 6 * sizeof(unsigned int), h->stream), "flag clear failed");
                unsigned int max_bits;
                std::memcpy(&max_bits, &xmax, sizeof max_bits);
                hipLaunchKernelGGL(k_verify_div, dim3(4096), dim3(256), 0, h->stream, h->d_flag,
                                   p.path_step_size, h->rcp_path_step, h->rcp_path_lo, 0.0f, max_bits);
                hipLaunchKernelGGL(k_verify_div, dim3(4096), dim3(256), 0, h->stream, h->d_flag + 3,
                                   s_range, h->rcp_range_hi, h->rcp_range_lo, (float)(p.s_steps - 1), max_bits);
                HIP_TRY(hipGetLastError(), "division check launch failed");
                unsigned int flags[6] = {1, 1, 1, 1, 1, 1};
                HIP_TRY(hipMemcpyAsync(flags, h->d_flag, sizeof flags, hipMemcpyDeviceToHost, h->stream), "flag copy failed");
                HIP_TRY(hipStreamSynchronize(h->stream), "division check failed");
                h->verified_path3 = (flags[0] == 0 ? 1 : 0) | (flags[1] == 0 ? 2 : 0) | (flags[2] == 0 ? 4 : 0);
                h->verified_range3 = (flags[3] == 0 ? 1 : 0) | (flags[4] == 0 ? 2 : 0) | (flags[5] == 0 ? 4 : 0);
                h->verified_c[0] = p.path_step_size;
                h->verified_c[1] = s_range;
                h->verified_xmax = xmax;
                h->verified_s_steps = p.s_steps;
            }
        }
        pick_div_forms(h);
    }
    // a re-parameterisation of an existing planner (tpl changes dt_start every cycle) only rebuilds the action table:
    // the launch is ordered before every later kernel of the stream, nothing here is read by the host
    if (resize) HIP_TRY(hipStreamSynchronize(h->stream), "building the action table failed");
    return TPLHIP_OK;
}

} // extern "C"

namespace {
// guard: record the event that protects the staging buffer against the NEXT upload.  A caller that waits for the whole
// stream before it returns (update_batch) does not need it -- and an event record between the upload and the first
// kernel of the plan holds that kernel back by several microseconds.
int upload_impl(tplhip_dp_lon* h, int n, const TplLonState* inits, const float* path, bool guard)
{
    if (!h || !inits || !path || n < 1) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    if (!h->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the planner");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    // A pending rollout of the back-to-back mode reads its OWN copy of path and initial states (enqueue_plan), so a new upload
    // only has to be ordered behind the previous plan on the stream -- unless the rollout buffers must grow, or the mode is off
    // and the rollout reads d_path / d_inits themselves.
    int rc = TPLHIP_OK;
    if (!h->overlap || n > h->cap_rollouts || !h->d_trajs) rc = wait_pending_forwards(h);
    if (rc) return rc;
    rc = ensure_rollouts(h, n);
    if (rc) return rc;
    h->n_rollouts = n;
    h->path_bounded = path_is_bounded(path, h->params.path_steps);
    // lon_planner.cu:334-338 and :365-368.  Through a page-locked staging buffer, so that the copies are true
    // asynchronous DMA requests and the host does not wait for them: the kernels that read them follow on the stream.
    const size_t path_b = sizeof(float) * 7 * (size_t)h->params.path_steps;
    const size_t init_b = sizeof(TplLonState) * (size_t)n;
    if (h->up_inflight) {
        HIP_TRY(hipEventSynchronize(h->ev_up), "upload failed");
        h->up_inflight = false;
    }
    if (path_b + init_b > h->up_cap) {
        if (h->h_up) (void)hipHostFree(h->h_up);
        h->h_up = nullptr; h->up_cap = 0;
        HIP_TRY(hipHostMalloc(&h->h_up, path_b + init_b, hipHostMallocDefault), "pinned buffer allocation failed");
        h->up_cap = path_b + init_b;
    }
    if (!h->ev_up) HIP_TRY(hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming), "event creation failed");
    std::memcpy(h->h_up, path, path_b);
    std::memcpy(h->h_up + path_b, inits, init_b);
#ifndef TPLHIP_UPLOAD_KERNEL
#define TPLHIP_UPLOAD_KERNEL 1
#endif
#if TPLHIP_UPLOAD_KERNEL
    // one launch instead of two copy commands (a hipMemcpyAsync costs the host ~6 us, a launch ~3, and the front of a plan
    // is bound by how fast the host submits): a few workgroups read the page-locked buffer through its device mapping
    {
        const unsigned words_a = (unsigned)(path_b / 4), words_b = (unsigned)(init_b / 4);
        const unsigned blocks = std::min(64u, (words_a + words_b + 255u) / 256u);
        hipLaunchKernelGGL(k_upload_words, dim3(blocks), dim3(256), 0, h->stream, reinterpret_cast<const unsigned*>(h->h_up),
                           reinterpret_cast<unsigned*>(h->d_path), words_a, reinterpret_cast<unsigned*>(h->d_inits), words_b);
        HIP_TRY(hipGetLastError(), "upload launch failed");
    }
#else
    HIP_TRY(hipMemcpyAsync(h->d_path, h->h_up, path_b, hipMemcpyHostToDevice, h->stream), "copying path failed");
    HIP_TRY(hipMemcpyAsync(h->d_inits, h->h_up + path_b, init_b, hipMemcpyHostToDevice, h->stream), "copying initial states failed");
#endif
    if (guard) {
        HIP_TRY(hipEventRecord(h->ev_up, h->stream), "event record failed");
        h->up_inflight = true;
    }
    return TPLHIP_OK;
}
} // namespace

extern "C" {

int tplhip_dp_lon_upload(tplhip_dp_lon* h, int n, const TplLonState* inits, const float* path)
{
    return upload_impl(h, n, inits, path, true);
}

int tplhip_dp_lon_plan_resident(tplhip_dp_lon* h, tplhip_dp_env* env)
{
    int rc = check_env(h, env);
    if (rc) return rc;
    if (h->n_rollouts < 1) return tplhip_fail(TPLHIP_ESTATE, "nothing uploaded");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    try_enable_overlap(h, env);
    return enqueue_plan(h, env, h->overlap);
}

int tplhip_dp_lon_sync(tplhip_dp_lon* h)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "lon planner pass failed");
    int rc = wait_pending_forwards(h);
    return rc ? rc : chain_check(h);
}

int tplhip_dp_lon_download(tplhip_dp_lon* h, TplLonState* out_trajs)
{
    if (!h || !out_trajs) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (h->n_rollouts < 1) return tplhip_fail(TPLHIP_ESTATE, "nothing to download");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    // the copy queues behind the rollout it reads (its own stream in the back-to-back mode, and that stream waited
    // for the sweeps), into a page-locked buffer: ONE host wait per plan
    const size_t n = (size_t)h->n_rollouts * h->params.t_steps;
    if (n > h->dl_cap) {
        if (h->h_dl) (void)hipHostFree(h->h_dl);
        h->h_dl = nullptr; h->dl_cap = 0;
        HIP_TRY(hipHostMalloc(&h->h_dl, sizeof(TplLonState) * n, hipHostMallocCoherent | hipHostMallocMapped), "pinned buffer allocation failed");
        h->dl_cap = n;
    }
    const bool on_fwd = h->fwd_stream && (h->fwd_pending[0] || h->fwd_pending[1]);
    hipStream_t cs = on_fwd ? h->fwd_stream : h->stream;
    HIP_TRY(hipMemcpyAsync(h->h_dl, h->d_trajs, sizeof(TplLonState) * n, hipMemcpyDeviceToHost, cs), "copying trajectory failed");
    HIP_TRY(hipStreamSynchronize(cs), "lon planner forward pass failed");
    if (on_fwd) h->fwd_pending[0] = h->fwd_pending[1] = false;
    h->up_inflight = false;                                  // the uploads were ordered before what just finished
    {
        int rc = chain_check(h);
        if (rc) return rc;
    }
    std::memcpy(out_trajs, h->h_dl, sizeof(TplLonState) * n);
    return TPLHIP_OK;
}

int tplhip_dp_lon_stage_download(tplhip_dp_lon* h, int slot)
{
    if (!h || slot < 0 || slot > 1) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    if (h->n_rollouts < 1) return tplhip_fail(TPLHIP_ESTATE, "nothing to download");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const size_t n = (size_t)h->n_rollouts * h->params.t_steps;
    if (n > h->stage_cap[slot]) {
        if (h->h_stage[slot]) (void)hipHostFree(h->h_stage[slot]);
        h->h_stage[slot] = nullptr; h->stage_cap[slot] = 0;
        HIP_TRY(hipHostMalloc(&h->h_stage[slot], sizeof(TplLonState) * n, hipHostMallocDefault), "pinned buffer allocation failed");
        h->stage_cap[slot] = n;
    }
    if (!h->stage_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&h->stage_ev[slot], hipEventDisableTiming), "event creation failed");
    // behind the rollout it copies: on the rollout stream in the back-to-back mode
    hipStream_t cs = (h->overlap && (h->fwd_pending[0] || h->fwd_pending[1])) ? h->fwd_stream : h->stream;
    HIP_TRY(hipMemcpyAsync(h->h_stage[slot], h->d_trajs, sizeof(TplLonState) * n, hipMemcpyDeviceToHost, cs),
            "copying trajectory failed");
    HIP_TRY(hipEventRecord(h->stage_ev[slot], cs), "event record failed");
    h->stage_busy[slot] = true;
    h->stage_n[slot] = n;
    return TPLHIP_OK;
}

int tplhip_dp_lon_collect_download(tplhip_dp_lon* h, int slot, TplLonState* out_trajs)
{
    if (!h || !out_trajs || slot < 0 || slot > 1) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    if (!h->stage_busy[slot]) return tplhip_fail(TPLHIP_ESTATE, "no download staged in slot %d", slot);
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    // the caller sizes out_trajs from the CURRENT rollout count and horizon: a slot staged under
    // other sizes (upload / reinit in between) cannot be handed out
    if (h->stage_n[slot] != (size_t)h->n_rollouts * h->params.t_steps) {
        h->stage_busy[slot] = false;
        return tplhip_fail(TPLHIP_ESTATE, "slot %d was staged with %zu states, the planner now holds %zu",
                           slot, h->stage_n[slot], (size_t)h->n_rollouts * h->params.t_steps);
    }
    HIP_TRY(hipEventSynchronize(h->stage_ev[slot]), "lon planner forward pass failed");
    {
        int rc = chain_check(h);
        if (rc) { h->stage_busy[slot] = false; return rc; }
    }
    std::memcpy(out_trajs, h->h_stage[slot], sizeof(TplLonState) * h->stage_n[slot]);
    h->stage_busy[slot] = false;
    return TPLHIP_OK;
}

int tplhip_dp_lon_update_batch(tplhip_dp_lon* h, int n, const TplLonState* inits,
                               const float* path, tplhip_dp_env* env, TplLonState* out_trajs)
{
    int rc = check_env(h, env);
    if (rc) return rc;
    if (!out_trajs) return tplhip_fail(TPLHIP_EINVAL, "null output");
    rc = upload_impl(h, n, inits, path, false);              // this call returns after the stream has drained
    if (rc) return rc;
    // the rollout writes the trajectories into page-locked host memory as well: the result is there when the stream has
    // drained -- no copy command behind the rollout (0.149 -> 0.146 ms per default-grid plan).  (In front of the plan the
    // upload is ONE small launch, see upload_impl; letting the plan's first kernel read the staging buffer itself was
    // slower -- ten blocks, now hundreds, fetching the same 5.6 KB over PCIe.)
    const size_t cnt = (size_t)n * h->params.t_steps;
    if (cnt > h->dl_cap) {
        if (h->h_dl) (void)hipHostFree(h->h_dl);
        h->h_dl = nullptr; h->dl_cap = 0;
        // (coherent + mapped, stated: the rollout kernel stores the trajectory and then the flag word through the device
        // mapping, and the host reads the trajectory as soon as it has SEEN the flag -- no runtime call in between)
        HIP_TRY(hipHostMalloc(&h->h_dl, sizeof(TplLonState) * cnt, hipHostMallocCoherent | hipHostMallocMapped), "pinned buffer allocation failed");
        h->dl_cap = cnt;
    }
    // one rollout = one workgroup: it publishes a sequence number in page-locked host memory behind its last store, and the
    // host spins on that word instead of asking the runtime to wait for the stream (a bounded spin: then the plain wait)
    if (!h->h_flag) {
        if (hipHostMalloc(&h->h_flag, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) *h->h_flag = 0;
        else { (void)hipGetLastError(); h->h_flag = nullptr; }               // no flag word: the plain stream wait
    }
    const bool spin = n == 1 && h->h_flag != nullptr;
    const unsigned int seq = ++h->flag_seq;
    rc = enqueue_plan(h, env, false, h->h_dl, spin ? h->h_flag : nullptr, seq);
    if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }          // the unguarded upload may still read the staging buffer
    bool seen = false;
    if (spin) {
        volatile unsigned int* f = h->h_flag;
        // bounded by TIME (2 ms: an order of magnitude above the slowest single-rollout plan), then the plain stream wait
        const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        for (long it = 0; !seen; ++it) {
            seen = __atomic_load_n(const_cast<unsigned int*>(f), __ATOMIC_ACQUIRE) == seq;
            if (seen) break;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#elif defined(__aarch64__)
            asm volatile("yield");
#else
            std::this_thread::yield();
#endif
            if ((it & 255) == 255 && std::chrono::steady_clock::now() > give_up) break;
        }
    }
    if (!seen) HIP_TRY(hipStreamSynchronize(h->stream), "lon planner forward pass failed");
    h->up_inflight = false;                                  // the uploads were ordered before what just finished
    rc = chain_check(h);
    if (rc) return rc;
    std::memcpy(out_trajs, h->h_dl, sizeof(TplLonState) * cnt);
    return TPLHIP_OK;
}

int tplhip_dp_lon_update(tplhip_dp_lon* h, const TplLonState* init, const float* path,
                         tplhip_dp_env* env, TplLonState* out_traj)
{
    return tplhip_dp_lon_update_batch(h, 1, init, path, env, out_traj);
}

int tplhip_dp_lon_reeval(tplhip_dp_lon* h, const TplLonState* traj, int len, const float* path,
                         tplhip_dp_env* env, TplLonState* out_traj)
{
    int rc = check_env(h, env);
    if (rc) return rc;
    if (!traj || !path || !out_traj) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    const TplDpLonParams& p = h->params;
    if (len != p.t_steps)                                     // lon_planner.cu:423-425
        return tplhip_fail(TPLHIP_ESTATE, "trajectory size does not match t_steps");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    if (h->cap_reeval < len) {
        (void)hipFree(h->d_reeval); h->d_reeval = nullptr; h->cap_reeval = 0;
        HIP_TRY(hipMalloc(&h->d_reeval, sizeof(TplLonState) * (size_t)len), "trajectory allocation failed");
        h->cap_reeval = len;
    }
    // a rollout of the back-to-back mode may still read d_path (and the environment's distance map) on its own stream
    rc = wait_pending_forwards(h);
    if (rc) return rc;
    // d_path now holds THIS path: a later plan_resident() plans on it, so the kernel choice follows it
    h->path_bounded = path_is_bounded(path, p.path_steps);
    HIP_TRY(hipMemcpyAsync(h->d_path, path, sizeof(float) * 7 * (size_t)p.path_steps,
                           hipMemcpyHostToDevice, h->stream), "copying path failed");
    HIP_TRY(hipMemcpyAsync(h->d_reeval, traj, sizeof(TplLonState) * (size_t)len,
                           hipMemcpyHostToDevice, h->stream), "copying trajectory to gpu failed");
    hipLaunchKernelGGL(k_dist_map_path, dim3(env->params.t_steps), dim3(256),
                       dist_slice_lds_bytes(env->params.s_steps), h->stream,
                       env->d_dist, env->d_occ, h->d_path, p.path_steps, p.path_step_size, env->params,
                       (size_t)0, (size_t)0, (size_t)0, h->rule);
    hipLaunchKernelGGL(k_reeval, dim3((len + 63) / 64), dim3(64), 0, h->stream,
                       h->d_reeval, h->d_tables, h->table_cells, h->d_path, env->d_dist, p, env->params, h->rule);
    HIP_TRY(hipGetLastError(), "reeval launch failed");
    HIP_TRY(hipMemcpyAsync(out_traj, h->d_reeval, sizeof(TplLonState) * (size_t)len,
                           hipMemcpyDeviceToHost, h->stream), "copying reevaluted trajectory to cpu failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "reeval pass failed");
    return chain_check(h);
}

int tplhip_dp_lon_copy_table(tplhip_dp_lon* h, int t, float* out)
{
    if (!h || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!h->have_params || !h->d_tables) return tplhip_fail(TPLHIP_ESTATE, "no value tables allocated");
    if (t < 0 || t >= h->params.t_steps) return tplhip_fail(TPLHIP_EINVAL, "stage %d out of range", t);
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    {
        int rc = chain_check(h);
        if (rc) return rc;
    }
    // the C-ABI hands out the reference's float4 cells (cost, constr, j, 0); the device keeps 12
    std::vector<Node> tmp(h->table_cells);
    HIP_TRY(hipMemcpy(tmp.data(), h->d_tables + (size_t)t * h->table_cells, sizeof(Node) * h->table_cells,
                      hipMemcpyDeviceToHost), "copying value table failed");
    for (size_t i = 0; i < h->table_cells; ++i) {
        out[4 * i + 0] = tmp[i].c; out[4 * i + 1] = tmp[i].g; out[4 * i + 2] = tmp[i].j; out[4 * i + 3] = 0.0f;
    }
    return TPLHIP_OK;
}

int tplhip_dp_lon_info(tplhip_dp_lon* h, char* buf, int n)
{
    if (!h || !buf || n < 1) return tplhip_fail(TPLHIP_EINVAL, "bad argument");
    const int kind = sweep_kind(h);
    const char* sweep = kind == 1 ? (h->rule.order != 0 ? "k_backward3o" : "k_backward3v") : (kind == 2 ? "k_backward3s" : "k_backward");
    std::snprintf(buf, (size_t)n,
                  "{\"sweep\": \"%s\", \"s_pad\": %d, \"rows_per_block\": %d, \"nblocks\": %d, \"kchunk\": %d, \"lds_bytes\": %zu, "
                  "\"div_verified_up_to\": %g, \"tile_v\": %d, \"tile_a\": %d, \"grid\": %d, \"big\": %d, "
                  "\"bounded_inputs\": %d, \"path_div3\": %d, \"range_div3\": %d, \"scenario_kernel\": \"%s\", \"fwd_overlap\": %d, \"plan_front\": \"%s\", "
                  "\"skew_pairs_per_a_row\": %.4f, \"tile_v3\": %d, \"rule\": {\"quant\": %d, \"lerp\": %d, \"fma\": %d, \"order\": %d}, "
                  "\"chained_sweep\": %d, \"chained_launches\": %llu, \"chained_failures\": %llu}",
                  sweep, h->s_pad, h->rows_per_block, h->nblocks, h->kchunk, h->lds_bytes,
                  (double)h->verified_xmax, h->tile_v, h->tile_a, h->grid, h->big ? 1 : 0,
                  (h->params_bounded && h->path_bounded && h->env_bounded) ? 1 : 0,
                  h->path_div3, h->range_div3, (h->v3_ok && use_v3(h) && h->sc_bounded && h->params_bounded) ? (h->rule.order != 0 ? "k_backward3o" : "k_backward3v") : "k_backward", h->overlap ? 1 : 0,
                  h->front_fused && !h->no_front ? "k_plan_front" : "three launches", (double)h->skew_q16 / 65536.0, h->tile_v3,
                  h->rule.quant, h->rule.lerp, h->rule.fma, h->rule.order,
                  chain_plan(h, true) ? (h->chain_mode == 1 ? 2 : 1) : 0, h->chain_launches, h->chain_failures);
    return TPLHIP_OK;
}

// The arithmetic rules of the HP-1 contract as run-time choices (FilterRule in dp_lon_math.h).  One helper applies a new
// rule: the handle is only changed AFTER the device is selected and pending rollouts -- which read h->rule -- have drained;
// it rebuilds the action table and re-picks the division forms.
namespace {
int apply_rule(tplhip_dp_lon* h, FilterRule r)
{
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    int rc = wait_pending_forwards(h);
    if (rc) return rc;
    h->rule = r;
    h->rule_set = !same_rule(r, kDefaultRule);
    h->rule_all_forms = r.quant == kDefaultRule.quant && r.lerp == kDefaultRule.lerp && r.order == kDefaultRule.order;
    if (h->have_params) pick_div_forms(h);
    h->tables_valid = false;
    if (h->have_params && h->d_acts) {
        rc = build_action_tables(h);
        if (rc) return rc;
    }
    return TPLHIP_OK;
}
} // namespace

// Chained sweep (SweepChain): -1 = the synchronous calls only (default), 0 = never, 1 = every plan of this handle.
int tplhip_dp_lon_set_chained_sweep(tplhip_dp_lon* h, int mode)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (mode < -1 || mode > 1) return tplhip_fail(TPLHIP_EINVAL, "chained sweep: -1 synchronous calls only, 0 never, 1 every plan");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const bool had = h->chain_mode != 0;
    h->chain_mode = mode;
    if (mode != 0 && !had && h->have_params && h->d_acts) {
        // the dependency lists are written with the traversal table
        int rc = wait_pending_forwards(h);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
        return build_action_tables(h);
    }
    return TPLHIP_OK;
}

// R2 / R3: the plain-fp32 sweeps are built for all nine (weight rule, blend form) pairs, so the choice of kernel does not change.
int tplhip_dp_lon_set_filter_rule(tplhip_dp_lon* h, int weight_quant, int lerp_form)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (weight_quant < 0 || weight_quant > 2 || lerp_form < 0 || lerp_form > 2)
        return tplhip_fail(TPLHIP_EINVAL, "filter rule: weight_quant in {0 none, 1 rint, 2 floor}, lerp_form in {0, 1, 2}");
    return apply_rule(h, FilterRule{weight_quant, lerp_form, h->rule.fma, h->rule.order});
}

// R1: 1 = the multiply-adds nvcc's default -fmad=true contracts, written out as fma() (FilterRule::fma; the sites are listed in
// the header of oracle/dp_lon_oracle.c) -- the DEFAULT since round 6, because that is how the reference's device code is built;
// 0 = one rounding per operation of the reference source.  A MODEL of nvcc's fusion until it has been checked against a dump or
// the SASS of the reference build.  Every kernel honours it: the plain-fp32 sweeps as template variants built
// with every division form for both settings, the general sweep, rollout, re-evaluation, stage prelude and distance map at
// run time (the instances that fold the default rule fold R1 = 1).
int tplhip_dp_lon_set_contraction(tplhip_dp_lon* h, int on)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (on != 0 && on != 1) return tplhip_fail(TPLHIP_EINVAL, "contraction: 0 (one rounding per operation) or 1 (fused multiply-adds)");
    return apply_rule(h, FilterRule{h->rule.quant, h->rule.lerp, on, h->rule.order});
}

// R3's ORDER / form (FilterRule::order): 0 = a -> v -> s (default), 1 = s -> v -> a, 2 / 3 = the eight-corner sum in fp32 /
// wide.  Rollout, re-evaluation, stage prelude and the general sweep honour it at run time; the large-grid sweep has its own
// form for it (k_backward3o, dp_sweep.hip: six raw rows parked, eight LDS taps per cell-action), built for the weight rules
// rint and floor -- un-quantised weights under another order take the general sweep.  The price is in DESIGN.md 0.
int tplhip_dp_lon_set_blend_order(tplhip_dp_lon* h, int order)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (order < 0 || order > 3)
        return tplhip_fail(TPLHIP_EINVAL, "blend order: 0 (a -> v -> s), 1 (s -> v -> a), 2 (eight-corner sum, fp32), 3 (eight-corner sum, wide)");
    return apply_rule(h, FilterRule{h->rule.quant, h->rule.lerp, h->rule.fma, order});
}

// Diagnostics, host only (no device needed): is a plain-fp32 sweep instance built for this arithmetic rule and this pair of
// division forms?  kind 1 = the two-rows-per-lane sweep (k_backward3v; k_backward3o when order != 0), 2 = one cell per lane
// (k_backward3s, both prologue forms).  What pick_div_forms / use_v3 may select must be a subset of what this reports.
int tplhip_dp_lon_sweep_built(int quant, int lerp, int fma, int order, int path_div, int range_div, int kind)
{
    const FilterRule r{quant, lerp, fma, order};
    SweepArgs a{};
    a.s_pad = 64;
    t_sweep_probe = true;
    bool ok = false;
    if (kind == 1) ok = (order == 0 || backward3o_built(r)) && launch_backward3v(r, path_div, range_div, a, dim3(1, 1), 0, nullptr);
    else if (kind == 2) ok = order == 0 && launch_backward3s(r, path_div, range_div, 3, false, a, dim3(1, 1), 0, nullptr)
                              && launch_backward3s(r, path_div, range_div, 3, true, a, dim3(1, 1), 0, nullptr);
    else if (kind == 3) { a.chain.stages = 2; ok = order == 0 && launch_backward3v(r, path_div, range_div, a, dim3(1, 2), 0, nullptr); }   // the chained form of k_backward3v
    t_sweep_probe = false;
    return ok ? 1 : 0;
}

int tplhip_dp_lon_timing_enable(tplhip_dp_lon* h, int enable)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    for (auto& ev : h->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    h->events.clear();
    h->sweep_ms = 0.0;
    h->sweep_launches = 0;
    h->timing = enable != 0;
    return TPLHIP_OK;
}

int tplhip_dp_lon_timing_get(tplhip_dp_lon* h, double* sweep_ms, int64_t* sweep_launches)
{
    if (!h || !sweep_ms || !sweep_launches) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    for (auto& ev : h->events) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second), "event query failed");
        h->sweep_ms += (double)ms;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    h->events.clear();
    *sweep_ms = h->sweep_ms;
    *sweep_launches = h->sweep_launches;
    return TPLHIP_OK;
}

} // extern "C"

// ---- scenario batches ----------------------------------------------------------

namespace {

void free_scenarios(tplhip_dp_lon* h)
{
    (void)hipFree(h->d_sc_occ); (void)hipFree(h->d_sc_path); (void)hipFree(h->d_sc_dist); (void)hipFree(h->d_sc_tables);
    (void)hipFree(h->d_sc_pre_vmax); (void)hipFree(h->d_sc_pre_sdist); (void)hipFree(h->d_sc_inits); (void)hipFree(h->d_sc_trajs);
    (void)hipFree(h->d_sc_best);
    h->d_sc_occ = h->d_sc_path = h->d_sc_dist = h->d_sc_pre_vmax = h->d_sc_pre_sdist = nullptr;
    h->d_sc_tables = nullptr; h->d_sc_inits = h->d_sc_trajs = nullptr; h->d_sc_best = nullptr;
    h->sc_n = h->sc_chunk = h->sc_paths = 0;
}

size_t record_bytes(int T) { return 16 + sizeof(TplLonState) * (size_t)T; }

} // namespace

void free_scenarios_fwd(tplhip_dp_lon* h) { free_scenarios(h); }

extern "C" {

size_t tplhip_dp_lon_record_bytes(int t_steps)
{
    return t_steps < 1 ? 0 : record_bytes(t_steps);
}

int tplhip_dp_lon_scenarios_upload(tplhip_dp_lon* h, int n, const float* occ_nlst, const TplDpEnvParams* ep_in,
                                   const TplLonState* inits, const float* paths, int n_paths, int chunk)
{
    if (!h || !ep_in || !paths || ((occ_nlst == nullptr) != (inits == nullptr))) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (!h->have_params) return tplhip_fail(TPLHIP_ESTATE, "reinit_buffers was never called on the planner");
    if (n < 1 || chunk < 1 || (n_paths != 1 && n_paths != n))
        return tplhip_fail(TPLHIP_EINVAL, "need n >= 1, chunk >= 1 and one path or one per scenario (n = %d, chunk = %d, n_paths = %d)", n, chunk, n_paths);
    if (ep_in->t_steps < 1 || ep_in->s_steps < 2 || ep_in->l_steps < 2)
        return tplhip_fail(TPLHIP_EINVAL, "environment needs t_steps >= 1, s_steps >= 2, l_steps >= 2");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    chunk = std::min(chunk, n);
    const TplDpLonParams& p = h->params;
    TplDpEnvParams ep = *ep_in;
    env_update_step_sizes(&ep);
    const size_t occ_elems = (size_t)ep.l_steps * ep.s_steps * ep.t_steps;
    const size_t dist_elems = (size_t)ep.t_steps * ep.s_steps;
    const size_t path_elems = (size_t)p.path_steps * 7;
    const size_t S = (size_t)p.s_steps, T = (size_t)p.t_steps;
    free_scenarios(h);
    HIP_TRY(hipMalloc(&h->d_sc_occ, sizeof(float) * occ_elems * n), "scenario occupancy allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_path, sizeof(float) * path_elems * n_paths), "scenario path allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_dist, sizeof(float) * dist_elems * chunk), "scenario distance map allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_tables, sizeof(Node) * h->table_cells * T * chunk), "scenario value table allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_pre_vmax, sizeof(float) * S * T * chunk), "scenario prelude allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_pre_sdist, sizeof(float) * S * T * chunk), "scenario prelude allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_inits, sizeof(TplLonState) * (size_t)n), "scenario state allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_trajs, sizeof(TplLonState) * T * n), "scenario trajectory allocation failed");
    HIP_TRY(hipMalloc(&h->d_sc_best, record_bytes(p.t_steps)), "plan record allocation failed");
    HIP_TRY(hipMemsetAsync(h->d_sc_tables, 0, sizeof(Node) * h->table_cells * T * chunk, h->stream), "value table clear failed");
    HIP_TRY(hipMemsetAsync(h->d_sc_trajs, 0, sizeof(TplLonState) * T * n, h->stream), "trajectory clear failed");
    if (occ_nlst) {
        HIP_TRY(hipMemcpyAsync(h->d_sc_occ, occ_nlst, sizeof(float) * occ_elems * n, hipMemcpyHostToDevice, h->stream), "copying occupancy maps failed");
        HIP_TRY(hipMemcpyAsync(h->d_sc_inits, inits, sizeof(TplLonState) * (size_t)n, hipMemcpyHostToDevice, h->stream), "copying initial states failed");
    } else {                                         // reserve only: the blocks follow (tplhip_dp_lon_scenarios_set_block)
        HIP_TRY(hipMemsetAsync(h->d_sc_occ, 0, sizeof(float) * occ_elems * n, h->stream), "occupancy clear failed");
        HIP_TRY(hipMemsetAsync(h->d_sc_inits, 0, sizeof(TplLonState) * (size_t)n, h->stream), "initial state clear failed");
    }
    HIP_TRY(hipMemcpyAsync(h->d_sc_path, paths, sizeof(float) * path_elems * n_paths, hipMemcpyHostToDevice, h->stream), "copying paths failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "scenario upload failed");
    h->sc_n = n; h->sc_chunk = chunk; h->sc_paths = n_paths; h->sc_ep = ep;
    {
        bool ok = true;
        for (int i = 0; i < n_paths && ok; ++i) ok = path_is_bounded(paths + path_elems * i, p.path_steps);
        const float vals[] = {ep.s_min, ep.s_max, ep.l_min, ep.l_max, ep.dt, ep.dt_start};
        for (float v : vals) ok = ok && std::isfinite(v) && std::fabs(v) < 1e6f;
        h->sc_bounded = ok && h->params_bounded;
    }
    h->sc_last_first = h->sc_last_count = 0;
    return TPLHIP_OK;
}

int tplhip_dp_lon_scenarios_set_block(tplhip_dp_lon* h, int first, int count, const float* occ_block, const TplLonState* inits)
{
    if (!h || !occ_block || !inits) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (h->sc_n < 1) return tplhip_fail(TPLHIP_ESTATE, "no scenarios uploaded");
    if (first < 0 || count < 1 || first > h->sc_n || count > h->sc_n - first)      // (no first + count: it overflows for huge arguments)
        return tplhip_fail(TPLHIP_EINVAL, "scenario block of %d from %d does not fit (%d scenarios)", count, first, h->sc_n);
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const TplDpEnvParams& ep = h->sc_ep;
    const size_t occ_elems = (size_t)ep.l_steps * ep.s_steps * ep.t_steps;
    HIP_TRY(hipMemcpyAsync(h->d_sc_occ + occ_elems * first, occ_block, sizeof(float) * occ_elems * count, hipMemcpyHostToDevice, h->stream),
            "copying occupancy maps failed");
    HIP_TRY(hipMemcpyAsync(h->d_sc_inits + first, inits, sizeof(TplLonState) * (size_t)count, hipMemcpyHostToDevice, h->stream),
            "copying initial states failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "scenario upload failed");
    return TPLHIP_OK;
}

int tplhip_dp_lon_scenarios_plan(tplhip_dp_lon* h, int first, int count, int id_base, int reset_best)
{
    if (!h) return tplhip_fail(TPLHIP_EINVAL, "null handle");
    if (h->sc_n < 1) return tplhip_fail(TPLHIP_ESTATE, "no scenarios uploaded");
    if (first < 0 || count < 1 || count > h->sc_chunk || first > h->sc_n || count > h->sc_n - first)
        return tplhip_fail(TPLHIP_EINVAL, "scenario range of %d from %d does not fit (%d scenarios, %d per launch)", count, first, h->sc_n, h->sc_chunk);
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    const TplDpLonParams& p = h->params;
    const TplDpEnvParams& ep = h->sc_ep;
    const int T = p.t_steps;
    const size_t S = (size_t)p.s_steps;
    const size_t occ_elems = (size_t)ep.l_steps * ep.s_steps * ep.t_steps;
    const size_t dist_elems = (size_t)ep.t_steps * ep.s_steps;
    const size_t path_stride = h->sc_paths > 1 ? (size_t)p.path_steps * 7 : 0;
    const float* paths = h->d_sc_path + path_stride * first;
    const size_t tab_stride = h->table_cells * (size_t)T, pre_stride = S * (size_t)T;

    hipLaunchKernelGGL(k_dist_map_path, dim3(ep.t_steps, count), dim3(256), dist_slice_lds_bytes(ep.s_steps), h->stream,
                       h->d_sc_dist, h->d_sc_occ + occ_elems * first, paths, p.path_steps, p.path_step_size, ep,
                       dist_elems, occ_elems, path_stride, h->rule);
    hipLaunchKernelGGL(k_stage_prelude, dim3(T, count), dim3(256), 0, h->stream,
                       h->d_sc_pre_vmax, h->d_sc_pre_sdist, h->d_pre_prog, paths, h->d_sc_dist, p, ep,
                       pre_stride, path_stride, dist_elems, h->rule);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing) {
        HIP_TRY(hipEventCreate(&e0), "event creation failed");
        HIP_TRY(hipEventCreate(&e1), "event creation failed");
        HIP_TRY(hipEventRecord(e0, h->stream), "event record failed");
    }
    const bool v3 = h->v3_ok && use_v3(h) && h->sc_bounded && h->params_bounded;
    for (int i = T - 1; i > 0; --i) {
        const int nx = std::min(T - 1, i + 1);
        Node* out = h->d_sc_tables + (size_t)i * h->table_cells;
        const Node* next = h->d_sc_tables + (size_t)nx * h->table_cells;
        const float* pv = h->d_sc_pre_vmax + (size_t)i * S;
        const float* ps = h->d_sc_pre_sdist + (size_t)i * S;
        if (i == T - 1) {                                            // rule R4
            hipLaunchKernelGGL((k_backward<true, 1, false>), dim3(h->grid, count), dim3(h->s_pad * h->rows_per_block), 0, h->stream,
                               out, out, h->d_acts, paths, pv, ps, h->d_pre_prog, p, h->s_pad, h->rows_per_block, h->nblocks,
                               h->tile_v, h->tile_a, tab_stride, pre_stride, path_stride, h->rule);
        } else if (v3 && [&] {
            SweepArgs a;
            a.next = next; a.out = out; a.acts = h->d_acts; a.path = paths; a.pre_vmax = pv; a.pre_sdist = ps;
            a.pre_prog = h->d_pre_prog; a.p = p; a.s_pad = h->vp_s_pad;
            a.rcp_path_hi = h->rcp_path_step; a.rcp_path_lo = h->path_aux;
            a.rcp_range_hi = h->rcp_range_hi; a.rcp_range_lo = h->range_aux;
            a.tile_v = h->tile_v3; a.tile_a = h->tile_a; a.skew_q16 = h->skew_q16; a.tile_map = h->d_v3_map;
            a.stride_tables = tab_stride; a.stride_pre = pre_stride; a.stride_path = path_stride;
            return launch_backward3v(h->rule, h->path_div3, h->range_div3, a, dim3(h->v3_grid, count),
                                     backward3v_lds_bytes(h->vp_s_pad, p.path_steps), h->stream);
        }()) {
            // (swept by the plain-fp32 kernel; no built instance for the pair -> the general sweep below, never a skipped stage)
        } else {
            auto go = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(h->grid, count), dim3(h->s_pad * h->rows_per_block), h->lds1_bytes, h->stream,
                                   next, out, h->d_acts, paths, pv, ps, h->d_pre_prog, p, h->s_pad, h->rows_per_block, h->nblocks,
                                   h->tile_v, h->tile_a, tab_stride, pre_stride, path_stride, h->rule);
            };
            if (h->rule.order != 0) go(k_backward<false, 1, false, true>);
            else if (h->use_buf) go(k_backward<false, 1, true>); else go(k_backward<false, 1, false>);
        }
    }
    if (h->timing) {
        HIP_TRY(hipEventRecord(e1, h->stream), "event record failed");
        h->events.emplace_back(e0, e1);
        h->sweep_launches += (T - 1);
    }
    {
        const size_t ring_b = sizeof(float) * 7 * FWD_RING;
        const size_t maps_b = sizeof(float) * ((size_t)p.path_steps * 7 + dist_elems);
        TplLonState* trajs = h->d_sc_trajs + (size_t)first * T;
        {
            const bool in_lds = ring_b + maps_b <= 60 * 1024, rt = h->rule_set;
            auto go = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(count), dim3(fwd_threads(h)), in_lds ? ring_b + maps_b : ring_b, h->stream,
                                   trajs, h->d_sc_inits + first, h->d_sc_tables, h->table_cells, paths, h->d_sc_dist, p, ep,
                                   tab_stride, path_stride, dist_elems, h->rule, (TplLonState*)nullptr, (unsigned int*)nullptr, 0u);
            };
            if (fwd_threads(h) == 64) {
                if (in_lds) { if (rt) go(k_forward_seq<true, true>); else go(k_forward_seq<true, false>); }
                else { if (rt) go(k_forward_seq<false, true>); else go(k_forward_seq<false, false>); }
            } else if (in_lds) { if (rt) go(k_forward<true, true>); else go(k_forward<true, false>); }
            else { if (rt) go(k_forward<false, true>); else go(k_forward<false, false>); }
        }
        hipLaunchKernelGGL(k_best_plan, dim3(1), dim3(256), 0, h->stream, trajs, count, id_base + first, T, h->d_sc_best, reset_best ? 1 : 0);
    }
    HIP_TRY(hipGetLastError(), "scenario plan launch failed");
    h->sc_last_first = first; h->sc_last_count = count;
    return TPLHIP_OK;
}

int tplhip_dp_lon_scenarios_download(tplhip_dp_lon* h, TplLonState* out_trajs)
{
    if (!h || !out_trajs) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (h->sc_n < 1) return tplhip_fail(TPLHIP_ESTATE, "no scenarios uploaded");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipMemcpyAsync(out_trajs, h->d_sc_trajs, sizeof(TplLonState) * (size_t)h->sc_n * h->params.t_steps,
                           hipMemcpyDeviceToHost, h->stream), "copying trajectories failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "scenario plans failed");
    return TPLHIP_OK;
}

int tplhip_dp_lon_scenarios_best(tplhip_dp_lon* h, void* record_out)
{
    if (!h || !record_out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (h->sc_n < 1) return tplhip_fail(TPLHIP_ESTATE, "no scenarios uploaded");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipMemcpyAsync(record_out, h->d_sc_best, record_bytes(h->params.t_steps), hipMemcpyDeviceToHost, h->stream),
            "copying the plan record failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "scenario plans failed");
    return TPLHIP_OK;
}

int tplhip_dp_lon_scenarios_best_device(tplhip_dp_lon* h, void** dev_record, size_t* bytes)
{
    if (!h || !dev_record || !bytes) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (h->sc_n < 1) return tplhip_fail(TPLHIP_ESTATE, "no scenarios uploaded");
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "scenario plans failed");      // the record is final when this returns
    *dev_record = h->d_sc_best;
    *bytes = record_bytes(h->params.t_steps);
    return TPLHIP_OK;
}

int tplhip_dp_lon_scenarios_copy_table(tplhip_dp_lon* h, int scenario, int t, float* out)
{
    if (!h || !out) return tplhip_fail(TPLHIP_EINVAL, "null argument");
    if (scenario < h->sc_last_first || scenario >= h->sc_last_first + h->sc_last_count)
        return tplhip_fail(TPLHIP_ESTATE, "scenario %d is not among those of the last launch [%d, %d)", scenario,
                           h->sc_last_first, h->sc_last_first + h->sc_last_count);
    if (t < 0 || t >= h->params.t_steps) return tplhip_fail(TPLHIP_EINVAL, "stage %d out of range", t);
    HIP_TRY(hipSetDevice(h->device), "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(h->stream), "stream sync failed");
    std::vector<Node> tmp(h->table_cells);
    const size_t slot = (size_t)(scenario - h->sc_last_first);
    HIP_TRY(hipMemcpy(tmp.data(), h->d_sc_tables + (slot * h->params.t_steps + (size_t)t) * h->table_cells,
                      sizeof(Node) * h->table_cells, hipMemcpyDeviceToHost), "copying value table failed");
    for (size_t i = 0; i < h->table_cells; ++i) {
        out[4 * i + 0] = tmp[i].c; out[4 * i + 1] = tmp[i].g; out[4 * i + 2] = tmp[i].j; out[4 * i + 3] = 0.0f;
    }
    return TPLHIP_OK;
}

} // extern "C"
