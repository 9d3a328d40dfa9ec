/*
 * tplhip_ilqr.h -- C-ABI of one generated batched iLQR solver library
 * (tpl_amd/models/libilqr_<problem>.so), HP-2 of the hot path.
 *
 * One library per optimal-control problem, like the reference's one CPython
 * module per problem (library/tpl/optim/genopt.py:486-523); every library
 * exports exactly these symbols.  They replace the CPython type "Optim" the
 * reference template defines (library/tpl/optim/templates/optim.c:1179-1994):
 * the Python shim tpl_amd.optim.solver.Optim rebuilds that type's attribute
 * surface on top of the host mirrors handed out by tplhip_ilqr_ptr().
 *
 * A handle holds `batch` independent problems of the same model.  batch = 1
 * reproduces the reference object; batch > 1 adds a leading axis to every
 * array.  All arithmetic is fp64 like the reference.  Nothing here falls back
 * to the CPU: update() without a usable HIP device fails with TPLHIP_EDEVICE.
 * (tplhip_ilqr_dynamics / _ct_dynamics evaluate one model routine on the
 * host, as the reference's Optim.dynamics does; that is the same generated
 * routine compiled for the host, not a second implementation.)
 */
#ifndef TPLHIP_ILQR_H
#define TPLHIP_ILQR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef TPLHIP_API
#define TPLHIP_API __attribute__((visibility("default")))
#endif

#ifndef TPLHIP_OK
#define TPLHIP_OK        0
#define TPLHIP_EINVAL   -1
#define TPLHIP_EDEVICE  -2
#define TPLHIP_ESTATE   -3
#endif
#define TPLHIP_EASSERT  -4   /* a shape check of the reference template failed (reference: AssertionError, optim.c:15-44, 1195-1200) */

#define TPLHIP_ILQR_H_MAX 300   /* optim.c:49: stages are clamped to [1, H_MAX-1] (:1726-1734) */

/* Per-problem arrays, optim.c:508-622 (Optim struct) in the order of its
 * __slots__ (:1736-1782).  Rows: R1 = horizon+1, R0 = horizon.            */
enum TplIlqrField {
    TPL_F_INT_STEP = 0,        /* [R1]        */
    TPL_F_X,                   /* [R1][X]     */
    TPL_F_U,                   /* [R0][U]     */
    TPL_F_NEXT_INT_STEP,       /* [R1]        */
    TPL_F_NEXT_X,              /* [R1][X]     */
    TPL_F_NEXT_U,              /* [R0][U]     */
    TPL_F_PREV_X,              /* [R1][X]     */
    TPL_F_PREV_U,              /* [R0][U]     */
    TPL_F_PREV_K,              /* [R0][U]     */
    TPL_F_FX,                  /* [R0][X][X]  */
    TPL_F_FU,                  /* [R0][X][U]  */
    TPL_F_LX,                  /* [R0][X]     */
    TPL_F_LU,                  /* [R0][U]     */
    TPL_F_LXX,                 /* [R0][X][X]  */
    TPL_F_LUU,                 /* [R0][U][U]  */
    TPL_F_LUX,                 /* [R0][U][X]  */
    TPL_F_G,                   /* [R0][U]     */
    TPL_F_K_FF,                /* k  [R0][U]  */
    TPL_F_K_FB,                /* K  [R0][U][X] */
    TPL_F_LAGRANGE_MULTIPLIER, /* [R0][C]     */
    TPL_F_BARRIER_WEIGHT,      /* [C]         */
    TPL_F_LG_MULT_LIMIT,       /* [C]         */
    TPL_F_U_MAX,               /* [R0][U]     */
    TPL_F_U_MIN,               /* [R0][U]     */
    TPL_F_COUNT
};

/* Per-problem scalars, optim.c:572-620 + members :1862-1882; stored as doubles. */
enum TplIlqrScalar {
    TPL_S_TRAJ_COSTS = 0,
    TPL_S_ITERATIONS,
    TPL_S_LG_ITERATIONS,
    TPL_S_RUNTIME,               /* ms; wall time of the last update() call */
    TPL_S_ALPHA,
    TPL_S_MU,
    TPL_S_MU_STEP,
    TPL_S_MIN_MU,
    TPL_S_TRAJECTORY_CHANGED,
    TPL_S_IMPROVED,
    TPL_S_TERMINATION_CONDITION,
    TPL_S_USE_QUADRATIC_TERMS,
    TPL_S_DT,
    TPL_S_MAX_ITERATIONS,
    TPL_S_MAX_LG_ITERATIONS,
    TPL_S_OPT_START,
    TPL_S_MIN_REL_COST_CHANGE,
    TPL_S_HORIZON,
    TPL_S_INTEGRATOR_TYPE,       /* 0 EULER, 1 HEUN, 2 RK4 (optim.c:491-495) */
    TPL_S_COUNT
};

typedef struct tplhip_ilqr tplhip_ilqr;

/* ---- static model description -------------------------------------------- */
TPLHIP_API const char* tplhip_ilqr_model_name(void);
TPLHIP_API void tplhip_ilqr_dims(int* X, int* U, int* C, int* n_params);
TPLHIP_API const char* tplhip_ilqr_param_name(int i);        /* Config order = Params.__slots__ order */
TPLHIP_API int tplhip_ilqr_param_is_array(int i);
TPLHIP_API const char* tplhip_ilqr_last_error(void);
/* sha1 of the problem text the library was generated from (genopt.Config.fingerprint) */
TPLHIP_API const char* tplhip_ilqr_fingerprint(void);

/* ---- lifetime --------------------------------------------------------------- */
/* t_cap: stages of storage per problem (horizon may be set to anything in
 * [1, t_cap]); the reference's static capacity is H_MAX-1 = 299.  Defaults as
 * pyOptimAlloc sets them (optim.c:1894-1921): dt .05, T 20 (clamped to t_cap),
 * minRelCostChange 1e-6, maxIterations 5, maxLgIterations 1,
 * useQuadraticTerms, lgMultLimit inf, barrierWeight 1, u in (-inf, inf) for
 * the first 20 stages and [0, 0] beyond (the reference only initialises T=20 rows). */
TPLHIP_API int tplhip_ilqr_create(int device, int batch, int t_cap, tplhip_ilqr** out);
TPLHIP_API int tplhip_ilqr_destroy(tplhip_ilqr* h);
TPLHIP_API int tplhip_ilqr_batch(tplhip_ilqr* h);
TPLHIP_API int tplhip_ilqr_t_cap(tplhip_ilqr* h);

/* ---- host mirrors (what the numpy views alias; optim.c:1314-1347) -------------- */
/* Field f of problem 0; problem b is at + b * (*problem_stride) doubles, rows
 * are dense ([row][...] with the trailing dims of the table above). */
TPLHIP_API double* tplhip_ilqr_ptr(tplhip_ilqr* h, int field, int64_t* problem_stride);
/* scalars[batch][TPL_S_COUNT] */
TPLHIP_API double* tplhip_ilqr_scalars(tplhip_ilqr* h);
/* Scalar parameter i: pointer to double[batch]. NULL if i is an array parameter. */
TPLHIP_API double* tplhip_ilqr_param_scalar_ptr(tplhip_ilqr* h, int i);
/* Array parameter i (copied in, like dynArrayFromPyObj optim.c:1361-1391).
 * per_problem = 0: one array shared by the whole batch, dims = its shape.
 * per_problem = 1: data is [batch] x dims. */
TPLHIP_API int tplhip_ilqr_set_param_array(tplhip_ilqr* h, int i, const double* data, int ndim,
                                           const int64_t* dims, int per_problem);
TPLHIP_API int tplhip_ilqr_get_param_array(tplhip_ilqr* h, int i, const double** data, int* ndim,
                                           int64_t* dims, int* per_problem);

/* ---- solver ------------------------------------------------------------------ */
/* Optim.update(), optim.c:1091-1160 via pyUpdate :1485-1494: uploads the
 * mirrors, runs the batched AL-iLQR on the GPU, downloads every field. */
TPLHIP_API int tplhip_ilqr_update(tplhip_ilqr* h);
/* Optim.shift(n), optim.c:1162-1177 (on the host mirrors). */
TPLHIP_API int tplhip_ilqr_shift(tplhip_ilqr* h, int amount);
/* Optim.dynamics / ct_dynamics, optim.c:1496-1652, for problem b's parameters. */
TPLHIP_API int tplhip_ilqr_dynamics(tplhip_ilqr* h, int b, const double* x, const double* u, int t, double dt, double* out);
TPLHIP_API int tplhip_ilqr_ct_dynamics(tplhip_ilqr* h, int b, const double* x, const double* u, int t, double dt, double* out);

/* ---- device-resident stepping, for measurement (bench.py) ----------------------- */
TPLHIP_API int tplhip_ilqr_upload(tplhip_ilqr* h);             /* mirrors -> device, also snapshots them */
TPLHIP_API int tplhip_ilqr_restore(tplhip_ilqr* h);            /* device-side copy snapshot -> working state (async) */
TPLHIP_API int tplhip_ilqr_run_resident(tplhip_ilqr* h);       /* one update() of the whole batch, async, no host traffic */
TPLHIP_API int tplhip_ilqr_sync(tplhip_ilqr* h);
TPLHIP_API int tplhip_ilqr_download(tplhip_ilqr* h);           /* device -> mirrors */
/* HIP-event time of the solver kernel launches since the last call with reset=1 */
/* Where the solver keeps its working set for the next launch, as a small JSON object (diagnostics / tests):
 * {"batch", "t_cap", "lds_traj": 1 when x, u, k, K, uMin, uMax are mirrored in LDS, "lds_arrays", "lds_dyn_bytes", ...,
 *  "form": the launch form of the last update, "scheduler": the instruction scheduler strategy the library was compiled under
 *  ("max-ilp" where its device listing passed the EXEC = 0 scan, else "default": tpl_amd/optim/genopt.py:compile_model)}.
 * Valid after the first update() / upload(). */
TPLHIP_API int tplhip_ilqr_info(tplhip_ilqr* h, char* buf, int n);
TPLHIP_API int tplhip_ilqr_timing(tplhip_ilqr* h, int reset, double* kernel_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* TPLHIP_ILQR_H */
