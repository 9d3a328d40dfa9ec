// dp_lon_internal.h -- types shared by the translation units of the DP (s,v,a) planner
// (dp_lon.hip: host driver + the general kernels; dp_sweep.hip: the large-grid sweep kernel).
#pragma once

#include <hip/hip_runtime.h>
#include "dp_lon_math.h"

namespace tplhip {

constexpr int N_BWD = 9;    // evalNode<9>,  lon_planner.cu:201
constexpr int N_FWD = 21;   // evalNode<21>, lon_planner.cu:220

// ActDesc::flags.  With a weight of exactly 0 the blend fma(q, 0, p * 1) returns p for every finite q (table values are
// finite and never -0: costs and constraint sums are >= +0, a jerk of zero is j_min + range * k/8 = +0), and with both
// a-rows clamped to the same row q IS p: in both cases the plain-fp32 sweeps request the a0 rows only and blend them
// with themselves -- the same bits for half the row loads of such an action (a third to a half of all actions on the
// benchmark grids: the jerk grid lands on the acceleration grid).
constexpr int ACT_SINGLE_A = 1;
// The row above -- (v + 1, a), the second row of this row's pair in the two-rows-per-lane sweeps; the row itself where v + 1 is
// past the grid -- reads for this action, as the lower pair of its source window, this row's upper pair under the same weight
// along a: "upper a-blend of row 0" IS "lower a-blend of row 1", one blend and two row loads less (dp_sweep.hip).
constexpr int ACT_SHARED_UP = 2;

// Per-(row, action) constants of the backward sweep.  Everything here depends
// only on (v, a, jerk_k, dt) -- never on the station or the stage.
// The first eight floats are the part a sweep needs in VECTOR registers (operands of plain fp32
// multiply / add instructions, which only run at full rate on vector operands -- tools/micro/valu_tp.hip);
// k_backward3v copies them to LDS once and reads them back as broadcasts.
struct alignas(16) ActDesc {
    float wa, iwa;      // quantised blend weight along a (rule R2) and 1 - wa (exact: multiples of 1/256)
    float wv, iwv;      // the same along v
    float j;            // jerk of this action
    float wj_cost;      // stateActionCost             lon_planner.cu:57-64
    float a_viol;       // (a_min - a') or (a' - a_max) if violated, else 0 (at most one is non-zero)
    float tn_v;         // successor velocity          lon_planner.cu:245
    double j_term;      // ((((1/6)*j)*dt)*dt)*dt      lon_planner.cu:244
    double tn_v_d;      // (double)tn_v, for the double-precision speed-limit term (:114-118)
    int r00, r10, r01, r11;   // next-stage rows (a*V + v) of the corners (a0,v0) (a1,v0) (a0,v1) (a1,v1)
    double a_term;      // ((0.5*a)*dt)*dt             lon_planner.cu:244   (the same for the nine actions of a row)
    float vdt;          // v*dt                                              (the same for the nine actions of a row)
    int flags;          // bit 0: the blend along a is trivial -- both a-rows are the same row (clamped) or wa == 0 -- see ACT_SINGLE_A;
                        // bit 1: ACT_SHARED_UP
};
static_assert(sizeof(ActDesc) == 80, "ActDesc must be 80 bytes");

// Traversal of the two-rows-per-lane sweeps (k_backward3v / k_backward3o): workgroup bx -> the row pair (a, v), (a, v + 1).
// XCD = bx & 7 owns a contiguous run of the row PAIRS of every a-slice (so any v_steps works; with an odd v_steps the last
// "pair" is row V-1 twice); inside its share the workgroups walk strips of tile_v / 2 pairs a-major in tiles of tile_a rows.
// Skew: the successor row of (a, v) under the action that leads to a' is v' = v + dt (a + a') / 2, so the source rows that
// read the SAME next-stage rows lie on lines v + a dt / 2 = const.  The pairs of an a-row are rotated (inside the XCD's share, a
// bijection per a-row) by that slope, skew_q16 / 65536 pairs per a-row: the workgroups resident on an XCD at one time -- a few
// pairs by several dozen a-rows -- then touch a band of v' rows as narrow as the strip itself instead of one widened by the
// a-extent (profiles/r04_dp_config4_skew.txt).  false: bx is past the XCD's share (the grid is sized for the widest one).
// ONE definition for k_fill_v3_map, which tabulates it per reinit (tplhip_dp_lon::d_v3_map -> SweepArgs::tile_map: the three
// integer divisions and the modulo are ~150 scalar instructions in front of a workgroup's first row request), and the kernels'
// fallback without a table.
__host__ __device__ inline bool v3_map_block(int bx, int V, int A, int tile_v, int tile_a, int skew_q16, int& ia, int& iv)
{
    const int xcd = bx & 7;
    const int NP = (V + 1) / 2;
    const int p_lo = (xcd * NP) >> 3, p_hi = ((xcd + 1) * NP) >> 3;
    const int vx = p_hi - p_lo, tv = tile_v / 2 > 1 ? tile_v / 2 : 1;
    const int q = bx >> 3;
    if (q >= vx * A) return false;
    const int full = vx / tv;
    int strip = q / (A * tv), w = tv, rem = q - strip * (A * tv);
    if (strip >= full) { strip = full; w = vx - full * tv; rem = q - full * (A * tv); }
    const int tile = rem / (tile_a * w);
    const int r = rem - tile * (tile_a * w);
    ia = tile * tile_a + r / w;
    int lp = strip * tv + (r - (r / w) * w);
    if (skew_q16) { lp -= (int)(((long long)ia * skew_q16) >> 16) % vx; if (lp < 0) lp += vx; }
    iv = 2 * (p_lo + lp);
    return true;
}
constexpr unsigned V3_MAP_NONE = 0xffffffffu;
// One entry of the tabulated traversal (k_fill_v3_map, dp_lon_kernels.h; launched behind every k_build_actions): where the
// workgroup works and the next-stage rows its FIRST requests go to -- action 0 of the pair: the lower-a rows (a0, v0) of row 0
// and (a0, v1) of both rows, ActDesc::r00 / r01.  One 16-byte scalar load.
struct alignas(16) V3Map {
    unsigned where;            // (a << 16) | v, V3_MAP_NONE past the XCD's share
    int r00, r01_lo, r01_up;   // acts[(a V + v) 9].r00, .r01 and acts[(a V + v + 1) 9].r01 (row v again where v + 1 is past the grid)
};
static_assert(sizeof(V3Map) == 16, "V3Map is one dwordx4");

// SEVERAL dependent stages in ONE launch of k_backward3v (grid.y = stages, the latest first): no kernel boundary between two
// sweeps, and the first workgroups of a stage run beside the last ones of the stage before.  Workgroups start in the order
// of their linear id on every XCD (x fastest, then y; id & 7 = XCD since grid.x is a multiple of 8), so every workgroup a
// waiting one depends on has already started -- and a wait that outlasts `budget` gives up and raises *err: the host discards
// the plan and sweeps stage by stage from then on; no wave can wait for ever.  What a workgroup waits for: the row pairs of the next stage
// its nine actions read (deps: up to CHAIN_DEPS flag indices per workgroup, -1 = none; k_fill_v3_map), each announced by
// its writer -- node stores written THROUGH the XCD's L2 (sc1), completed, then flags[stage][pair] = epoch, all at agent
// scope.  A reader's L2 cannot hold an older copy of such a row: rows are whole cache lines (the host chains only where a
// row is a multiple of 128 bytes), every stage table is written once per plan, and it is first read after its flag.
constexpr int CHAIN_DEPS = 64;
struct SweepChain {
    int stages = 0;            // 0: a plain launch (blockIdx.y = scenario)
    unsigned epoch = 0;        // this launch's value of a set flag
    unsigned wait_epoch = 0;   // what a waiting workgroup looks for: `epoch` (TPLHIP_DEBUG=chain_stall: a value nobody writes -- tests)
    unsigned* flags = nullptr; // [t_steps][pairs], pairs = a_steps * ((v_steps + 1) / 2)
    int first = 0;             // stage index of blockIdx.y = 0
    int pairs = 0;
    const int* deps = nullptr; // [grid.x][CHAIN_DEPS]
    unsigned* err = nullptr;   // device word: non-zero once a wait gave up (what waiting workgroups look at)
    unsigned* err_host = nullptr;   // the same news in page-locked host memory (what the host looks at: no copy per plan)
    size_t table_cells = 0;    // nodes per stage table
    long long budget = 0;      // s_memtime ticks a workgroup may wait
};

// Everything a sweep launch needs besides the tables (filled by the host driver in dp_lon.hip).
struct SweepArgs {
    const Node* next;
    Node* out;
    const ActDesc* acts;
    const float* path;
    const float* pre_vmax;     // this stage, [s_steps]
    const float* pre_sdist;    // this stage, [s_steps]
    const double* pre_prog;    // [s_steps]
    TplDpLonParams p;
    int s_pad;
    float rcp_path_hi, rcp_path_lo;      // 1/path_step_size as a two-float number (division forms 5 and 0: _lo = the divisor)
    float rcp_range_hi, rcp_range_lo;    // 1/(s_max - s_min) likewise
    int tile_v, tile_a;
    int skew_q16;              // k_backward3v: rotation of the row pairs of an a-row, pairs per a-row in 16.16 fixed point (0: none)
    const V3Map* tile_map;     // k_backward3v / 3o: v3_map_block tabulated per workgroup (V3Map); null: computed
    // scenario batch (blockIdx.y): element strides between consecutive scenarios, 0 = shared
    size_t stride_tables, stride_pre, stride_path;
    // chained sweep (k_backward3v, one scenario): blockIdx.y = c sweeps stage (this launch's first stage - c); see SweepChain
    SweepChain chain;
};

// true (on the calling thread only): the launchers below answer "is this combination built?" and launch nothing
// (tplhip_dp_lon_sweep_built; tests/test_capi.py walks every (rule, division forms) pair the host can select)
extern thread_local bool t_sweep_probe;

// dp_sweep.hip.  path_div / range_div: the form of the two divisions by run-time constants -- 4 = power-of-two divisor
// (one multiplication), 3 = two-float reciprocal (two plain fp32 instructions), 5 = reciprocal and one Newton step on the
// quotient (three), each only after k_verify_div proved it equal to IEEE division on the whole range; 0 = IEEE division.
// For forms 5 and 0 rcp_*_lo carries the divisor itself.  rule: R2 / R3 (FilterRule); a rule other than the default
// is built with the pairs (5, 5) and (0, 0) only.  Returns false when the combination is not built.
bool launch_backward3v(FilterRule rule, int path_div, int range_div, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);
size_t backward3v_lds_bytes(int s_pad, int path_steps);
// k_backward3o: the same sweep for rule.order != 0 (six raw rows parked, eight LDS taps per cell-action); launch_backward3v
// dispatches to it and sizes its LDS block itself.  Built for the weight rules rint and floor.
size_t backward3o_lds_bytes(int s_pad, int path_steps);
bool backward3o_built(FilterRule rule);
// k_backward3s: the same instruction stream, one cell per lane and kc (= 3) actions per barrier round, for grids
// that leave the chip mostly empty (s_pad <= 256; grid = 8 * rows of the widest XCD share, as k_backward's)
size_t backward3s_lds_bytes(int kc, int s_pad, int path_steps);
bool launch_backward3s(FilterRule rule, int path_div, int range_div, int kc, bool first_early, const SweepArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);

} // namespace tplhip
