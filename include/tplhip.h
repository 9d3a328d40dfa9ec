/*
 * tplhip.h -- C-ABI of libtplhip.so, the MI355X-native hot path for tpl.
 *
 * This is the drop-in boundary: plain C structs, pointers and sizes, no torch
 * or C++ types.  Every entry point names the reference interface it replaces
 * (paths relative to the reference checkout, file:line).
 *
 * HP-1  DP (s,v,a) longitudinal planner   -> tplcpp.DynProgLonPlanner & friends
 *       (library/src/python/bindings_dp_lon.cu:7-137, library/src/dyn_prog/lon_planner.cu)
 * HP-2  genopt iLQR solver                -> genopt<sha1>.Optim
 *       (library/tpl/optim/templates/optim.c:1179-1994)     [declared in tplhip_ilqr.h]
 *
 * All functions return 0 on success and a negative TPLHIP_E* code on failure;
 * the message of the most recent failure on the calling thread is returned by
 * tplhip_last_error().  Nothing in here falls back to a CPU implementation:
 * without a usable HIP device every compute entry point fails with
 * TPLHIP_EDEVICE.
 */
#ifndef TPLHIP_H
#define TPLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPLHIP_API __attribute__((visibility("default")))

#define TPLHIP_OK        0
#define TPLHIP_EINVAL   -1   /* bad argument / shape (reference: ValueError)            */
#define TPLHIP_EDEVICE  -2   /* HIP runtime failure (reference: RuntimeError from CUDA) */
#define TPLHIP_ESTATE   -3   /* call order / size mismatch (reference: RuntimeError)    */

/* ---------------------------------------------------------------- PODs -- */

/* DynProgLonPlannerParams, library/include/tplcpp/dyn_prog/lon_planner.cuh:7-53.
 * Same fields, same order, same defaults (see tplhip_dp_lon_params_default). */
typedef struct TplDpLonParams {
    float s_min, s_max;
    float v_min, v_max;
    float a_min, a_max;
    float j_min, j_max;
    int32_t t_steps, s_steps, v_steps, a_steps;
    float dt_start, dt;
    float _s_step_size, _v_step_size, _a_step_size; /* derived, updateStepSizes :45-50 */
    float time_gap, gap_min;
    float w_progress, w_a, w_j, w_snap, w_safety_dist;
    float path_step_size;
    int32_t path_steps;
    float width_veh, length_veh;
} TplDpLonParams;

/* LonState, lon_planner.cuh:55-67 */
typedef struct TplLonState {
    float t, s, v, a, j, cost, constr;
} TplLonState;

/* PathState, library/include/tplcpp/dyn_prog/common.cuh:98-110.
 * A path is float[path_steps][7] in exactly this field order. */
typedef struct TplPathState {
    float x, y, s, l, k, v_max, distance;
} TplPathState;

/* DynProgEnvParams, library/include/tplcpp/dyn_prog/env.cuh:11-44 */
typedef struct TplDpEnvParams {
    float t_min, t_max;
    float l_min, l_max;
    float s_min, s_max;
    float ds_max;
    float dir_min, dir_max;
    float dt, dt_start;
    int32_t t_steps, l_steps, s_steps, dir_steps;
    float scale_objects, dilation;
    float _l_step_size, _s_step_size;               /* derived, updateStepSizes :39-43 */
} TplDpEnvParams;

typedef struct tplhip_dp_env tplhip_dp_env;   /* replaces DynProgEnvironment (the slice HP-1 touches) */
typedef struct tplhip_dp_lon tplhip_dp_lon;   /* replaces DynProgLonPlanner */

/* ------------------------------------------------------------- general -- */

TPLHIP_API const char* tplhip_last_error(void);
/* tplcpp.cuda_get_device_count, bindings_dp_common.cu:38-45 (0 on error) */
TPLHIP_API int tplhip_device_count(void);
/* tplcpp.cuda_set_device, bindings_dp_common.cu:47-49: handles created with device < 0 use this device
 * (default: the last one, as tpl/planning/__init__.py:2-7 selects) */
TPLHIP_API int tplhip_set_device(int idx);
TPLHIP_API const char* tplhip_version(void);

TPLHIP_API void tplhip_dp_lon_params_default(TplDpLonParams* p);   /* lon_planner.cuh:9-43 defaults */
TPLHIP_API void tplhip_dp_env_params_default(TplDpEnvParams* p);   /* env.cuh:13-37 defaults        */

/* ------------------------------------------------- host-only helpers ---- */

/* lonDynamics, lon_planner.cu:240-250 (bound as tplcpp.lon_dynamics, bindings_dp_lon.cu:85) */
TPLHIP_API void tplhip_lon_dynamics(const TplLonState* x0, float j, float dt, TplLonState* out);
/* LonTraj::state(t), lon_planner.cu:252-260 with InterpVars utils.hpp:53-86 */
TPLHIP_API int tplhip_lon_traj_state(const TplLonState* states, int n, float t, TplLonState* out);
/* the array overload of LonTraj.state (bindings_dp_lon.cu:93-108): out = double[m][4] = (s, v, a, j) at ts[m] */
TPLHIP_API int tplhip_lon_traj_state_rows(const TplLonState* states, int n, const double* ts, int m, double* out);

/* -------------------------------------------- DP environment (HP-1 a11) -- */

/* DynProgEnvironment(), env.cu:282-285.  device < 0 selects the last device
 * like tpl/planning/__init__.py:2-7 does. */
TPLHIP_API int tplhip_dp_env_create(int device, tplhip_dp_env** out);
TPLHIP_API int tplhip_dp_env_destroy(tplhip_dp_env* env);
/* DynProgEnvironment::reinitBuffers, env.cu:304-350: buffers are re-allocated
 * only when t_steps/l_steps/s_steps change.  Step sizes are written back. */
TPLHIP_API int tplhip_dp_env_reinit(tplhip_dp_env* env, TplDpEnvParams* ps);
/* Contents of EnvironmentGpu::occ_map as DynProgEnvironment::getOccMap hands
 * them out (bindings_dp_env.cu:84-90): float[l_steps][s_steps][t_steps], t
 * fastest.  In the reference this table is produced by the OpenGL occupancy
 * pipeline (env.cu:381-437), which is outside the hot path. */
TPLHIP_API int tplhip_dp_env_set_occ_map(tplhip_dp_env* env, const float* occ_lst);
TPLHIP_API int tplhip_dp_env_get_occ_map(tplhip_dp_env* env, float* occ_lst);
/* Result of updateDistMapPath (env.cu:131-158) for the path of the most recent
 * planner call: float[t_steps][s_steps], s fastest. */
TPLHIP_API int tplhip_dp_env_get_dist_map_path(tplhip_dp_env* env, float* dist_ts);

/* ------------------------ occupancy pipeline (env.cu:352-437) -- */

/* DynProgEnvironment::setRefLine (env.cu:376-379) + the RefLine constructor
 * (utils.hpp:184-214): ref_line is float64[n][9] = {x, y, heading, s, k, v_max,
 * d_left, d_right, semantic}; the bounding-box centre is subtracted and kept. */
TPLHIP_API int tplhip_dp_env_set_ref_line(tplhip_dp_env* env, const double* ref_line, int n, float step_size);
/* RefLine::x_offset / y_offset (bindings_dp_common.cu:34-35) */
TPLHIP_API int tplhip_dp_env_ref_line_offset(tplhip_dp_env* env, double* x_offset, double* y_offset);
/* DynProgEnvironment::insertGeometry (env.cu:352-374): geom is float64[n][3] =
 * (x, y, t) triangle-soup vertices; queued until the next update().  TPLHIP_ESTATE
 * ("cannot insert dyn object points with ref_line of 0 steps") without a reference line. */
TPLHIP_API int tplhip_dp_env_insert_geometry(tplhip_dp_env* env, const double* geom, int n, int stationary);
/* DynProgEnvironment::update (env.cu:381-437): rasterise the queued triangles
 * (replaces occupancy_renderer.cu:275-320), occupancyToFrenet, closeIntersections,
 * updateDistMapLon.  TPLHIP_ESTATE with the reference's message when the reference
 * line is shorter than s_max. */
TPLHIP_API int tplhip_dp_env_update(tplhip_dp_env* env);
/* Only the last stage of update(), updateDistMapLon (env.cu:95-129), on the Frenet occupancy the
 * environment holds (from update() or tplhip_dp_env_set_occ_map).  Not in the reference's bindings:
 * it lets a table dumped with get_occ_map() from the CUDA build drive the planners that read the
 * longitudinal distance map, independent of the rasterisation stages (tests/cycle_golden_format.py). */
TPLHIP_API int tplhip_dp_env_update_dist_maps(tplhip_dp_env* env);
/* get_occ_map_cartesian (bindings_dp_env.cu:76-83): uint32[size][size], size from
 * tplhip_dp_env_cart_size (1 before the first update, like the reference's 1x1 frame buffer) */
TPLHIP_API int tplhip_dp_env_cart_size(tplhip_dp_env* env, int* size_pixels);
TPLHIP_API int tplhip_dp_env_get_occ_map_cartesian(tplhip_dp_env* env, uint32_t* out);
/* getDistMapLon (env.cu:474-489): float[l_steps][s_steps][t_steps][2] =
 * (distance to the next occupied cell ahead, behind) */
TPLHIP_API int tplhip_dp_env_get_dist_map_lon(tplhip_dp_env* env, float* out);
/* The same two tables with their time slices below each other -- float[t_steps * l_steps][s_steps] and
 * float[t_steps * l_steps][s_steps][2] -- which is how the environment plugin publishes them every cycle
 * (dp_env.py:185-188 does np.vstack(np.moveaxis(get_occ_map(), 2, 0))): transposed on the device, one copy out. */
TPLHIP_API int tplhip_dp_env_get_occ_map_slices(tplhip_dp_env* env, float* out);
TPLHIP_API int tplhip_dp_env_get_dist_map_lon_slices(tplhip_dp_env* env, float* out);
/* DynProgEnvironment::getDist (env.cu:439-450): filtered occupancy at (t, s, l) */
TPLHIP_API int tplhip_dp_env_get_dist(tplhip_dp_env* env, float t, float s, float l, float* out);

/* ----------------------------------------------------- DP planner (HP-1) -- */

/* DynProgLonPlanner(), bindings_dp_lon.cu:120-121 */
TPLHIP_API int tplhip_dp_lon_create(int device, tplhip_dp_lon** out);
/* ~DynProgLonPlanner / clearBuffers, lon_planner.cu:262-282 */
TPLHIP_API int tplhip_dp_lon_destroy(tplhip_dp_lon* h);
/* DynProgLonPlanner::reinitBuffers, lon_planner.cu:284-326.  Writes the
 * derived step sizes back into *ps.  Unlike the reference, device tables are
 * re-allocated only when a size field (t/s/v/a_steps, path_steps) changed. */
TPLHIP_API int tplhip_dp_lon_reinit(tplhip_dp_lon* h, TplDpLonParams* ps);
/* DynProgLonPlanner::update, lon_planner.cu:328-401 (binding :123-129).
 * path: float[path_steps][7] host memory; out_traj: TplLonState[t_steps]. */
TPLHIP_API int tplhip_dp_lon_update(tplhip_dp_lon* h, const TplLonState* init,
                         const float* path, tplhip_dp_env* env,
                         TplLonState* out_traj);
/* Same backward sweep, n forward rollouts (one wave each) from n initial
 * states; out_trajs: TplLonState[n][t_steps].  New functionality. */
TPLHIP_API int tplhip_dp_lon_update_batch(tplhip_dp_lon* h, int n, const TplLonState* inits,
                               const float* path, tplhip_dp_env* env,
                               TplLonState* out_trajs);
/* DynProgLonPlanner::reevalTraj, lon_planner.cu:403-470 (binding :130-136).
 * len != t_steps -> TPLHIP_ESTATE like the reference's runtime_error (:423-425).
 * Uses the value tables of the most recent update. */
TPLHIP_API int tplhip_dp_lon_reeval(tplhip_dp_lon* h, const TplLonState* traj, int len,
                         const float* path, tplhip_dp_env* env,
                         TplLonState* out_traj);
/* Debug / parity: copy value table of stage t, float[a_steps][v_steps][s_steps][4]
 * = (cost, constr, j*, 0) as surf3Dwrite stores it (lon_planner.cu:203-208). */
TPLHIP_API int tplhip_dp_lon_copy_table(tplhip_dp_lon* h, int t, float* out);

/* ---- enqueue-only planning: throughput callers and measurement (bench.py) --------------- */

/* Upload path + initial states; afterwards tplhip_dp_lon_plan_resident enqueues one complete plan (dist map,
 * t_steps-1 backward sweeps, n forward rollouts) on the planner's stream without touching host memory.  Neither call
 * waits for the device: a caller may replan with NEW inputs back to back -- upload, plan_resident(any environment),
 * stage_download, upload, ... -- and a plan's rollout, which runs on a stream of its own behind the sweeps, reads its
 * own copy of path, start states and distance map (since round 6; until then upload waited on the host for the
 * previous rollout).  upload only waits when the rollout buffers must grow (more start states than ever before), and
 * for the staging buffer of the upload before last.  Every plan is bit-identical to tplhip_dp_lon_update on the same
 * inputs (tests/test_dp_gpu.py::test_replanning_with_new_path_start_and_environment_back_to_back). */
TPLHIP_API int tplhip_dp_lon_upload(tplhip_dp_lon* h, int n, const TplLonState* inits,
                         const float* path);
TPLHIP_API int tplhip_dp_lon_plan_resident(tplhip_dp_lon* h, tplhip_dp_env* env);
TPLHIP_API int tplhip_dp_lon_sync(tplhip_dp_lon* h);
TPLHIP_API int tplhip_dp_lon_download(tplhip_dp_lon* h, TplLonState* out_trajs);
/* Pipelined form of download for callers that keep enqueuing plans: stage_download enqueues the
 * copy of the plan just enqueued into one of two pinned host slots (behind it on the stream, so a
 * following plan_resident does not disturb it); collect_download waits for THAT copy only and
 * hands the trajectories out -- the next plan keeps the GPU busy meanwhile. */
TPLHIP_API int tplhip_dp_lon_stage_download(tplhip_dp_lon* h, int slot);
TPLHIP_API int tplhip_dp_lon_collect_download(tplhip_dp_lon* h, int slot, TplLonState* out_trajs);
/* Launch geometry and code path of the sweep kernel as a small JSON object (diagnostics). */
TPLHIP_API int tplhip_dp_lon_info(tplhip_dp_lon* h, char* buf, int n);
/* The two choices of the HP-1 arithmetic contract that stand in for the reference's TEXTURE UNIT -- undocumented
 * hardware arithmetic behind tex3D (library/include/tplcpp/dyn_prog/arr_tex_surf.cuh:142-149), frozen here by reading,
 * not by running -- as a run-time parameter of the planner, so that a dump of the CUDA build
 * which disagrees (tests/test_dp_reference_golden.py names the variant) is answered by this call:
 *   weight_quant  1 = filter weight rint(a * 256) / 256 (default), 2 = floor(a * 256) / 256, 0 = not quantised
 *   lerp_form     0 = fma(q, w, p * (1 - w)) (default), 1 = p * (1 - w) + q * w (every operation rounded), 2 = p + w * (q - p)
 * The plain-fp32 sweeps are built for all nine combinations: the choice of kernel does not depend on the rule. */
TPLHIP_API int tplhip_dp_lon_set_filter_rule(tplhip_dp_lon* h, int weight_quant, int lerp_form);
/* Rule R1 of the same contract: 1 (DEFAULT since round 6) = the multiply-adds nvcc contracts by default in device code
 * (-fmad=true: library/CMakeLists.txt:89-99 passes nothing else to nvcc; its -ffast-math reaches the host compiler only),
 * written out as fma() calls -- lonDynamics (lon_planner.cu:240-250), interpPath (common.cuh:124-136), the grid points
 * (lon_planner.cu:196-199, env.cu:144), s_dist and the cost sums of evalNode (lon_planner.cu:40-65, 84, 110, 148);
 * 0 = one IEEE rounding per operation of the reference source (the default until round 5).  Deterministic (no compiler
 * decides): oracle/dp_lon_oracle.c under dpo_set_rule(., ., on, .) is the same arithmetic.  Every kernel honours it; the
 * plain-fp32 sweeps are built with every division form for BOTH settings, so neither the choice of kernel nor its speed
 * depends on it.  R1 = 1 is a MODEL of nvcc's fusion -- the sites and the operand order (fma(a, b, c*d), left to right) are
 * placed by hand on both sides, so bit parity is proven against this repository's restatement, not against an nvcc build --
 * until it has been checked against a dump or the SASS of the reference build ("Pinning HP-1").  The host
 * helpers tplhip_lon_dynamics / tplhip_lon_traj_state are the HOST instantiation of the reference's function (compiled for
 * baseline x86-64, which has no fused multiply-add): one rounding per operation whatever this switch says. */
TPLHIP_API int tplhip_dp_lon_set_contraction(tplhip_dp_lon* h, int on);
/* Rule R3's ORDER and form (round 5).  The texture unit behind tex3D (lon_planner.cu:26-38, arr_tex_surf.cuh:142-149)
 * blends eight texels; the default contract freezes three sequential 1-D blends along a, then v, then s.  The hardware's
 * documentation gives the symmetric weighted sum of the eight corners, and a sequential implementation would as plausibly
 * start with x (= s).  order: 0 = a -> v -> s (default), 1 = s -> v -> a, 2 = sum of w_c T_c over the eight corners with
 * exact weight products and an fp32 accumulator in the order of the documented formula (corner = s + 2 v + 4 a), 3 = the
 * same sum accumulated wide and rounded once (lerp_form does not apply to 2 and 3).  oracle/dp_lon_oracle.c under
 * dpo_set_rule(., ., ., order) is the same arithmetic; tests/test_dp_reference_golden.py lists the variants a dump is scored
 * against.  Honoured by every kernel of the planner.  The large-grid sweep has its own form for another order
 * (tplhip_dp_lon_info -> "sweep": "k_backward3o": weight rules rint / floor, row buffers + path <= 64 KB of LDS, on every
 * grid); un-quantised weights under another order, and inputs the plain-fp32 sweeps refuse, take the general sweep
 * ("k_backward"). */
TPLHIP_API int tplhip_dp_lon_set_blend_order(tplhip_dp_lon* h, int order);
/* Chained sweep (round 6): the stages of one plan behind the terminal one -- the reference's loop of lonBackwardStep launches,
 * lon_planner.cu:347-361, with a device-wide wait after each -- in ONE launch of the two-rows-per-lane sweep.  A workgroup of
 * stage t waits for the row pairs of stage t + 1 its nine actions read (flags at agent scope; the writers store their nodes
 * through the XCD's L2); no kernel boundary between two stages: 256x128x64 one plan at a time 35-40 -> 34-37 us per stage
 * sweep (by box: what varies is the cost of a boundary), 27.4 -> 29.2 k sweeps/s; same bits (tests/test_dp_gpu.py::test_chained_sweep...).  Two planners kept busy side by
 * side LOSE 4 % with it (a waiting workgroup holds a slot the other plan could use), hence
 *   mode -1 (default): the synchronous calls (tplhip_dp_lon_update / _update_batch: one plan at a time by construction);
 *   mode  1: every plan of this handle, tplhip_dp_lon_plan_resident included;   mode 0: never.
 * Taken where the sweep is "k_backward3v" under the default filter rule and blend order (either setting of R1), a table row
 * is a multiple of 128 bytes (s_steps % 32 == 0), t_steps >= 4, a stage fills the chip at least once (workgroups x
 * wavefronts >= 32 x compute units: below that the chained launch loses) and a stage table is at most 96 MB
 * (tplhip_dp_lon_info -> "chained_sweep": 0 no / 1 synchronous calls / 2 every plan, "chained_launches", "chained_failures").  No wait is
 * unbounded: a workgroup that has waited 1 s gives up and flags it, the call that hands out that plan's result fails with
 * TPLHIP_EDEVICE ("plan again") and the handle sweeps stage by stage from then on.  It cannot happen while workgroups start
 * in the order of their ids, which every use so far has shown. */
TPLHIP_API int tplhip_dp_lon_set_chained_sweep(tplhip_dp_lon* h, int mode);
/* Diagnostics, host only (works without a GPU): 1 if a plain-fp32 sweep instance is built for this arithmetic rule and this
 * pair of division forms (4 = one multiplication, 3 = two-float reciprocal, 5 = reciprocal + Newton step, 0 = IEEE division),
 * else 0.  kind 1 = the two-rows-per-lane sweep (k_backward3v; k_backward3o when order != 0), 2 = one cell per lane
 * (k_backward3s), 3 = the chained form of k_backward3v (tplhip_dp_lon_set_chained_sweep: the default filter rule and blend order
 * under either R1).  The planner only ever selects built combinations (tests/test_capi.py walks them all); should that ever
 * drift, a stage without an instance is swept by the general kernel, never skipped. */
TPLHIP_API int tplhip_dp_lon_sweep_built(int weight_quant, int lerp_form, int contraction, int order, int path_div, int range_div, int kind);
/* HIP-event timing of the backward sweeps (events recorded on the planner's
 * stream around the loop of launches).  enable=1 starts accumulating; get
 * returns the summed ms and the number of sweep launches since enable. */
TPLHIP_API int tplhip_dp_lon_timing_enable(tplhip_dp_lon* h, int enable);
TPLHIP_API int tplhip_dp_lon_timing_get(tplhip_dp_lon* h, double* sweep_ms, int64_t* sweep_launches);

/* ---- scenario batches: Monte-Carlo rollouts of the planner (new; SURVEY 8e, BASELINE configs[4]) ----
 *
 * Nothing in the reference plans more than one scenario at a time (DynProgLonPlanner::update,
 * lon_planner.cu:328-401, is one plan with a host sync per stage).  Here n independent scenarios --
 * n occupancy maps and initial states, one path each or one shared path; planner parameters shared --
 * stay resident in HBM, and `count` of them are swept per launch (the scenario is a grid dimension of
 * every kernel: own value tables, distance map and stage prelude per scenario), which is what fills the
 * chip on the small grids tpl really runs.  Every scenario's plan is bit-identical to
 * tplhip_dp_lon_update on the same inputs.  After the rollouts of a launch the lexicographically
 * smallest (constraint violation, cost, id) plan is reduced ON THE DEVICE into a fixed-size record:
 * the message of the one all-gather a multi-GPU run needs. */

/* header of the best-plan record; TplLonState[t_steps] follows (tplhip_dp_lon_record_bytes in all) */
typedef struct TplDpPlanRecord {
    float constr;       /* max of states[1..t_steps-2].constr: the validity measure of the Python caller
                           (planning/dyn_prog/poly_lat_dp_lon_planner.py:99-101); all states when t_steps <= 2 */
    float cost;         /* states[0].cost = value of the whole plan */
    int32_t id;         /* id_base + index of the scenario; 0x7fffffff = no plan yet */
    int32_t t_steps;
} TplDpPlanRecord;

TPLHIP_API size_t tplhip_dp_lon_record_bytes(int t_steps);
/* occ_nlst: float[n][l_steps][s_steps][t_steps] (the layout of tplhip_dp_env_get_occ_map, per scenario);
 * inits: TplLonState[n]; paths: float[n_paths][path_steps][7] with n_paths = 1 (shared) or n.
 * chunk = scenarios swept per launch (value tables for that many are allocated: chunk x t_steps x cells x 12 B). */
TPLHIP_API int tplhip_dp_lon_scenarios_upload(tplhip_dp_lon* h, int n, const float* occ_nlst, const TplDpEnvParams* ep,
                                              const TplLonState* inits, const float* paths, int n_paths, int chunk);
/* Block-wise form of the upload, for scenario sets whose occupancy maps the caller would rather not hold in host memory
 * at once (1 024 scenarios of 31 x 201 x 50 floats are 1.25 GB): _scenarios_upload with occ_nlst == NULL and
 * inits == NULL reserves everything for n scenarios (maps zeroed), then every block [first, first + count) is handed
 * over with _scenarios_set_block (occ_block: float[count][l][s][t], inits: TplLonState[count]); synchronous, the
 * caller may reuse its buffer afterwards. */
TPLHIP_API int tplhip_dp_lon_scenarios_set_block(tplhip_dp_lon* h, int first, int count, const float* occ_block,
                                                 const TplLonState* inits);
/* Enqueue the complete plans of scenarios [first, first + count), count <= chunk: distance maps, stage preludes,
 * t_steps-1 batched sweeps, the forward rollouts and the reduction into the record (reset_best != 0 starts a new
 * record, 0 keeps the better of the record and this launch).  Scenario i carries the id id_base + i. */
TPLHIP_API int tplhip_dp_lon_scenarios_plan(tplhip_dp_lon* h, int first, int count, int id_base, int reset_best);
/* all n trajectories (TplLonState[n][t_steps]; scenarios never planned read as zeros); synchronises */
TPLHIP_API int tplhip_dp_lon_scenarios_download(tplhip_dp_lon* h, TplLonState* out_trajs);
/* the record on the host (record_bytes bytes), or its device address for a collective that sends it from HBM */
TPLHIP_API int tplhip_dp_lon_scenarios_best(tplhip_dp_lon* h, void* record_out);
TPLHIP_API int tplhip_dp_lon_scenarios_best_device(tplhip_dp_lon* h, void** dev_record, size_t* bytes);
/* parity: stage t of the value table of a scenario of the most recent launch, layout of tplhip_dp_lon_copy_table */
TPLHIP_API int tplhip_dp_lon_scenarios_copy_table(tplhip_dp_lon* h, int scenario, int t, float* out);

/* ------------- lateral polynomial planner (poly_lat_planner.cu) -- */

/* PolyLatPlannerParams, poly_lat_planner.cuh:11-59 (field order kept) */
typedef struct TplPolyLatParams {
    float l_min, l_max, s_min, s_max;
    int s_steps;
    float l_dst_min, l_dst_max, s_dst_min, s_dst_max;
    int l_dst_steps, s_dst_steps;
    float l_trg;
    float w_l, w_k, w_dl, w_ddl, w_dddl, w_right, w_len;
    float k_abs_max, a_lat_abs_max;
    float width_veh, length_veh;
    float _s_step_size, _l_dst_step_size, _s_dst_step_size;
} TplPolyLatParams;

/* PolyLatTrajPoint, poly_lat_planner.cuh:61-76 (72 bytes with natural alignment) */
typedef struct TplPolyLatTrajPoint {
    float t, l, dl, ddl, dddl, s, v;
    double x, y;
    float heading;
    double distance;
    float k;
} TplPolyLatTrajPoint;

/* PolyQuintic, poly_interp.cuh:34-61 */
typedef struct TplPolyQuintic {
    double x0, x1, d;
    double c[6];
} TplPolyQuintic;

typedef struct tplhip_poly_lat tplhip_poly_lat;

TPLHIP_API void tplhip_poly_lat_params_default(TplPolyLatParams* p);
/* PolyQuintic(x0, y0, dy0, ddy0, x1, y1, dy1, ddy1) and f / df / ddf / dddf (order = 0..3),
 * poly_interp.cu:96-168, 384-416.  Host only. */
TPLHIP_API void tplhip_poly_quintic_init(TplPolyQuintic* q, double x0, double y0, double dy0, double ddy0,
                                         double x1, double y1, double dy1, double ddy1);
TPLHIP_API double tplhip_poly_quintic_eval(const TplPolyQuintic* q, int order, double x);

TPLHIP_API int tplhip_poly_lat_create(int device, tplhip_poly_lat** out);
TPLHIP_API int tplhip_poly_lat_destroy(tplhip_poly_lat* h);
/* PolyLatPlanner::reinitBuffers, poly_lat_planner.cu:330-352; step sizes are written back */
TPLHIP_API int tplhip_poly_lat_reinit(tplhip_poly_lat* h, TplPolyLatParams* ps);
/* PolyLatPlanner::update, poly_lat_planner.cu:354-486: evaluates the l_dst x s_dst lattice of
 * quintic paths on the environment (reference line + occupancy of its last update), selects
 * one, and returns its s_steps points (x, y include the reference line offsets), the quintic
 * and the selected lattice indices. */
TPLHIP_API int tplhip_poly_lat_update(tplhip_poly_lat* h, const TplPolyLatTrajPoint* start, tplhip_dp_env* env,
                                      TplPolyLatTrajPoint* out_points, TplPolyQuintic* out_poly,
                                      int* best_idx_l_dst, int* best_idx_s_dst);
/* Debug / parity: the node lattice after the last update, float[l_dst][s_dst][s][4] =
 * (l, v, collision station or 10000, local cost); element s = 0 of every path holds
 * (.., .., min collision station, path cost) like the reference's path_nodes. */
TPLHIP_API int tplhip_poly_lat_copy_nodes(tplhip_poly_lat* h, float* out);
/* PolyLatTraj::updateTimeDistCurv / lerp, poly_lat_planner.cu:255-320.  Host only.  The
 * reference reads one element past the end for the LAST point's curvature (undefined);
 * here that curvature is 0. */
TPLHIP_API int tplhip_poly_lat_traj_update(TplPolyLatTrajPoint* points, int n);
TPLHIP_API int tplhip_poly_lat_traj_lerp(const TplPolyLatTrajPoint* points, int n, double distance,
                                         TplPolyLatTrajPoint* out);
/* the array overload of PolyLatTraj.lerp and PolyLatTraj.as_numpy (bindings_dp_poly_lat.cu:55-97): rows of 12 doubles
 * (t, l, dl, ddl, dddl, s, v, x, y, heading, distance, k) */
TPLHIP_API int tplhip_poly_lat_traj_lerp_rows(const TplPolyLatTrajPoint* points, int n, const double* distances, int m,
                                              double* out);
TPLHIP_API int tplhip_poly_lat_traj_rows(const TplPolyLatTrajPoint* points, int n, double* out);

/* --------- lateral/longitudinal DP planner (lat_lon_planner.cu) -- */

/* DynProgLatLonPlannerParams, lat_lon_planner.cuh:9-62 (field order kept) */
typedef struct TplLatLonParams {
    float s_min, s_max, ds_min, ds_max, l_min, l_max;
    float dds_min, dds_max, dl_min, dl_max;
    int t_steps, s_steps, ds_steps, l_steps;
    float dt, dt_start, dt_smooth_traj;
    float dds_start, w_dds_start, angle_start, w_angle_start;
    float l_trg;
    float w_progress, w_dds, w_ddds, w_l, w_dl, w_ddl, w_safety_dist, w_xing_slow;
    float slope_abs_max;
    float w_lat_dist, d_lat_comf;
    float time_gap, gap_min;
    float t_st_min, t_st_max, s_st, w_spatio_temporal;
    float width_veh, length_veh;
} TplLatLonParams;

/* LatLonState, lat_lon_planner.cuh:81-103 (48 bytes) */
typedef struct TplLatLonState {
    float t;
    float s, ds, dds, ddds;
    float l, dl, ddl, dddl;
    float cost, constr;
    uint8_t flags_constr;            /* LatLonConstr: 1 occupancy, 2 velocity, 4 angle */
} TplLatLonState;

/* LatLonCartState, lat_lon_planner.cuh:64-79 (88 bytes) */
typedef struct TplLatLonCartState {
    double t, distance, x, y, v, a, heading, k, cost, constr;
    uint8_t flags_constr;
} TplLatLonCartState;

typedef struct tplhip_lat_lon tplhip_lat_lon;

TPLHIP_API void tplhip_lat_lon_params_default(TplLatLonParams* p);
/* dp_lat_lon_planner_cu::dynamics, lat_lon_planner.cu:10-22.  Host. */
TPLHIP_API void tplhip_lat_lon_dynamics(const TplLatLonState* x0, float dds, float dl, float dt, TplLatLonState* out);
/* LatLonTraj::state / lerp, LatLonCartTraj::lerp, lat_lon_planner.cu:432-486.  Host. */
TPLHIP_API int tplhip_lat_lon_traj_state(const TplLatLonState* states, int n, float t, TplLatLonState* out);
TPLHIP_API int tplhip_lat_lon_traj_lerp(const TplLatLonState* states, int n, float t, TplLatLonState* out);
TPLHIP_API int tplhip_lat_lon_cart_lerp(const TplLatLonCartState* states, int n, float t, TplLatLonCartState* out);

TPLHIP_API int tplhip_lat_lon_create(int device, tplhip_lat_lon** out);
TPLHIP_API int tplhip_lat_lon_destroy(tplhip_lat_lon* h);
/* DynProgLatLonPlanner::reinitBuffers, lat_lon_planner.cu:515-577 */
TPLHIP_API int tplhip_lat_lon_reinit(tplhip_lat_lon* h, const TplLatLonParams* ps, int force);
/* updateTrajDp, :579-643: states[0] is the start state; the t_steps states of the DP plan come
 * back in states[0..t_steps-1].  Needs an environment whose update() ran. */
TPLHIP_API int tplhip_lat_lon_update_traj_dp(tplhip_lat_lon* h, tplhip_dp_env* env, TplLatLonState* states);
/* The texture unit's frozen arithmetic (tplhip_dp_lon_set_filter_rule / _set_blend_order above: weight rule, blend form, blend
 * order / eight-corner sums) for THIS planner's filtered fetches: the forward pass' lerp_cached_scaled of the value tables
 * (lat_lon_planner.cu:206-273 with interp = true, arr_tex_surf.cuh:160-167), interpDistMapLon (env.cu:236-252) and the
 * reference-line texture (common.cuh:93-95).  The backward pass point-samples (no filter).  Default (1, 0, 0). */
TPLHIP_API int tplhip_lat_lon_set_filter_rule(tplhip_lat_lon* h, int weight_quant, int lerp_form, int blend_order);
/* updateTrajSmooth, :645-760 (host: resampling + two LQR smoothing passes, utils.hpp:278-348).
 * smooth0 = trajSmooth.states[0]; out holds up to cap states, *n_out their number. */
TPLHIP_API int tplhip_lat_lon_update_traj_smooth(tplhip_lat_lon* h, const TplLatLonState* traj_dp, int n_dp,
                                                 const TplLatLonState* smooth0, TplLatLonState* out, int cap, int* n_out);
/* frenetToCartesian, :768-817 (host), on the reference line the environment holds */
TPLHIP_API int tplhip_lat_lon_frenet_to_cartesian(tplhip_dp_env* env, const TplLatLonState* traj, int n,
                                                  TplLatLonCartState* out);
/* reevalTraj, :819-859 */
TPLHIP_API int tplhip_lat_lon_reeval_traj(tplhip_lat_lon* h, tplhip_dp_env* env, const TplLatLonState* traj, int n,
                                          TplLatLonState* out);
/* copyBackwardPassTex / queryBackwardPassTex, :861-901: float[l_steps][ds_steps][s_steps][4] =
 * (cost, constr, dds*, dl*) of stage idx_t; the query is the point-sampled lookup */
TPLHIP_API int tplhip_lat_lon_copy_backward_pass_tex(tplhip_lat_lon* h, int idx_t, float* out);
TPLHIP_API int tplhip_lat_lon_query_backward_pass_tex(tplhip_lat_lon* h, int idx_t, float s, float ds, float l, float* out4);

/* --------- host helpers of the path/velocity-decomposition glue -- */

/* Projection, library/include/tplcpp/utils.hpp:240-256 (what tplcpp.project returns) */
typedef struct TplProjection {
    int64_t start, end;
    double alpha;
    int64_t index;
    double point[2];
    double distance, arc_len, angle;
    double tangent[2];
    int32_t in_bounds;
} TplProjection;

/* RefLine(data, len, step_size, no_offset), utils.hpp:184-214: copies double[n][9] reference-line points into `out`
 * with the bounding-box centre subtracted (unless no_offset) and returns that centre */
TPLHIP_API int tplhip_ref_line_center(const double* points, int n, int no_offset, double* out, double* x_offset,
                                      double* y_offset);
/* RefLine::lerp = interpRefLine, utils.hpp:148-173: the point at station s, out = double[9] */
TPLHIP_API int tplhip_ref_line_lerp(const double* points, int n, float step_size, float s, double* out);

/* tplcpp.gen_prediction_geometry, utils.cpp:576-693 (binding :1007-1048): pred = double[n_pred][5] (t, x, y, heading,
 * v), hull = double[n_hull][2], path = double[n_path][2], ts = double[n_ts]; writes the triangle-soup vertices
 * double[*n_out][3] = (x, y, t) that insert_geometry takes.  out == NULL only counts; TPLHIP_EINVAL when cap is too
 * small (then *n_out still holds the count). */
TPLHIP_API int tplhip_gen_prediction_geometry(const double* pred, int n_pred, const double* hull, int n_hull,
                                              const double* path, int n_path, const double* ts, int n_ts,
                                              double station_step_size, double expansion_rate, double sweep_length,
                                              double* out, int cap, int* n_out);
/* tplcpp.point_in_polygon, utils.cpp:25-65 (binding :780-804): out[k] = 1 where points[k] lies inside the polygon
 * double[n][2] */
TPLHIP_API int tplhip_point_in_polygon(const double* points, int m, const double* polygon, int n, uint8_t* out);
/* tplcpp.intersect_polygons, utils.cpp:67-158: separating-axis test of two convex polygons; *out = 1 when they overlap */
TPLHIP_API int tplhip_intersect_polygons(const double* poly0, int l0, const double* poly1, int l1, int* out);
/* tplcpp.convex_hull, utils.cpp:160-255: counter-clockwise hull from the lowest point; out = double[<= n][2] */
TPLHIP_API int tplhip_convex_hull(const double* points, int n, double* out, int* n_out);
/* tplcpp.smooth_path, utils.cpp:693-776: LQR smoothing of a line strip sampled every ds; out = double[n][2] */
TPLHIP_API int tplhip_smooth_path(const double* path, int n, double ds, double w_v, double w_a, double w_j, int closed,
                                  double* out);
/* tplcpp.lerp, utils.cpp:931-990: ys = double[m][d] sampled at the sorted xs[m], evaluated at vals[n] into
 * out = double[n][d]; angle: short-way-round interpolation; clip_alpha: hold the end values outside [xs[0], xs[m-1]] */
TPLHIP_API int tplhip_lerp(const double* vals, int n, const double* xs, int m, const double* ys, int d, int angle,
                           int clip_alpha, double* out);
/* tplcpp.project, utils.cpp:257-408: m positions (double[m][2]) on the line strip points (x, y in the first two
 * of `stride` doubles per point); closed != 0 treats it as a loop */
TPLHIP_API int tplhip_path_project(const double* points, int stride, int n, int closed, const double* positions, int m,
                                   TplProjection* out);
/* tplcpp.resample, utils.cpp:410-548: out = double[steps][5] = (x, y, alpha, prev index, next index); TPLHIP_ESTATE
 * with the reference's messages where it throws */
TPLHIP_API int tplhip_path_resample(const double* points, int stride, int n, double sampling_dist, int steps,
                                    int start_index, int closed, double* out, int* n_out);
/* rampify_profile, library/tpl/planning/utils.py:5-65: out = double[n][2] = (v, a) */
TPLHIP_API int tplhip_rampify_velocity_profile(int has_v0, double v0, int has_a0, double a0, const double* lim_v, int n,
                                               double a_min, double a_max, double j_min, double j_max, double v_min,
                                               double step, double* out);
/* rampify_profile, library/tpl/planning/path_vel_decomp/path_optim.py:11-59: out = double[n_path] */
TPLHIP_API int tplhip_rampify_lateral_profile(double step, int horizon, double evasion_sharpness, double proj_distance,
                                              const double* path_v, int path_stride, int n_path, double gap,
                                              const double* lower, const double* upper, double* out);

#ifdef __cplusplus
}
#endif
#endif /* TPLHIP_H */
