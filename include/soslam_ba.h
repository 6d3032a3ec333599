/*
 * soslam_ba.h - C ABI of the MI355X-native bundle-adjustment backend.
 *
 * The reference has no FFI for this path: its boundary is the C++ class API
 *   BundleAdjuster::Optimize(unsigned start, unsigned end)   /root/reference/src/bundle_adjuster.h:18
 * whose body gathers flat double[6] poses / double[3] points, adds one
 * AutoDiffCostFunction<ReprojectionError,4,6,3> + HuberLoss(1.0) per observation,
 * fixes the first pose and calls ceres::Solve (/root/reference/src/bundle_adjuster.cpp:39-118).
 * Everything from "problem assembled" to "poses and points updated" is replaced by the
 * entry points below; the host shim in stereo_orb_slam_amd/host/ keeps the class API.
 *
 * Conventions (all preserved from the reference):
 *   pose    double[6] = angle-axis(3) | translation(3) of the WORLD->CAMERA transform
 *           (/root/reference/src/bundle_adjuster.cpp:66-69, /root/reference/src/reprojection_error.h:19-24)
 *   point   double[3] world coordinates
 *   obs     float[4]  u_l v_l u_r v_r (/root/reference/src/observation.h:12-15; sigma is never read)
 *   proj    double[12] row-major 3x4 (/root/reference/src/reprojection_error.h:27-33,65-66)
 * Plain pointers and sizes only; every array is caller-allocated; the library never frees
 * caller memory.  A handle owns one HIP stream's worth of device state and is not thread-safe;
 * distinct handles are independent.  Every function returns an int status (0 = ok); on error
 * caller arrays are left unmodified.
 */
#ifndef SOSLAM_BA_H
#define SOSLAM_BA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SOSLAM_OK = 0,
    SOSLAM_ERR_INVALID_ARGUMENT = 1,
    SOSLAM_ERR_HIP = 2,              /* HIP runtime error (message via soslam_last_error) */
    SOSLAM_ERR_NO_DEVICE = 3,        /* no gfx950 device / extension cannot run */
    SOSLAM_ERR_NON_FINITE = 4,       /* non-finite cost at the initial point */
    SOSLAM_ERR_LINEAR_SOLVER = 5,    /* Cholesky breakdown / point block not positive definite */
    SOSLAM_ERR_COMM = 6,             /* all-reduce callback failed */
    SOSLAM_ERR_STATE = 7             /* call sequence error (no problem / no state set) */
};

/*
 * Reduced camera system solvers (the reference: SPARSE_SCHUR + CHOLMOD, /root/reference/src/bundle_adjuster.cpp:26-29).
 *   DENSE_CHOLESKY  block-sparse S expanded to dense, blocked right-looking Cholesky
 *   PCG             conjugate gradients on the block-sparse S; preconditioner = block-band Cholesky factor when
 *                   S fits a band of <= 15 block diagonals (camera chains), block-Jacobi otherwise
 *   BAND_CHOLESKY   direct block-band Cholesky (requires the <= 15 block-diagonal band)
 *   AUTO            BAND_CHOLESKY if the band fits, else DENSE_CHOLESKY up to 1200 camera dof, else PCG
 */
enum { SOSLAM_SOLVER_AUTO = 0, SOSLAM_SOLVER_DENSE_CHOLESKY = 1, SOSLAM_SOLVER_PCG = 2, SOSLAM_SOLVER_BAND_CHOLESKY = 3 };

enum {
    SOSLAM_TERM_MAX_ITERATIONS = 0,
    SOSLAM_TERM_PARAMETER_TOLERANCE = 1,
    SOSLAM_TERM_FUNCTION_TOLERANCE = 2,
    SOSLAM_TERM_GRADIENT_TOLERANCE = 3,
    SOSLAM_TERM_MIN_RADIUS = 4,
    SOSLAM_TERM_INVALID_STEPS = 5,
    SOSLAM_TERM_TIME = 6
};

/*
 * Solver options.  Defaults reproduce the reference's effective configuration
 * (/root/reference/src/bundle_adjuster.cpp:14-36, /root/reference/src/params.h:34-47) with Ceres'
 * defaults for everything the reference leaves unset, except
 * max_solver_time_seconds, which defaults to 0 = off (the reference's 1.0 s wall-clock cap makes the
 * iterate machine-dependent; set it to 1.0 to get the stock behaviour).
 */
typedef struct soslam_ba_options {
    int32_t max_iterations;          /* 50 */
    int32_t check_termination;       /* 1; 0 = run exactly max_iterations LM iterations */
    int32_t linear_solver;           /* SOSLAM_SOLVER_* */
    int32_t pcg_max_iterations;      /* 500 */
    double  pcg_tolerance;           /* relative residual |r|/|b| at which the reduced solve stops, 1e-10 */
    double  huber_delta;             /* 1.0 */
    double  lower_bound;             /* -1e4, applied to every point coordinate */
    double  upper_bound;             /* +1e4 */
    double  initial_radius;          /* 1e4 */
    double  max_radius;              /* 1e16 */
    double  min_radius;              /* 1e-32 */
    double  min_relative_decrease;   /* 1e-3 */
    double  min_lm_diagonal;         /* 1e-6 */
    double  max_lm_diagonal;         /* 1e32 */
    double  parameter_tolerance;     /* 1e-8 */
    double  function_tolerance;      /* 1e-16 */
    double  gradient_tolerance;      /* 1e-16 */
    double  max_solver_time_seconds; /* 0 = off */
    int32_t jacobi_scaling;          /* 1 */
    int32_t verbose;                 /* 1 = one line per iteration on stdout, like minimizer_progress_to_stdout */
    int32_t device;                  /* HIP device ordinal; -1 = current device */
    int32_t profile_stages;          /* 1 = bracket every stage with HIP events (summary.stage_ms) */
    void*   stream;                  /* hipStream_t to run on; NULL = the library creates its own */
} soslam_ba_options;

#define SOSLAM_BA_NUM_STAGES 8
enum {
    SOSLAM_STAGE_LINEARIZE = 0,   /* residual + Jacobian + loss + camera-block reduction (ba_linearize) */
    SOSLAM_STAGE_POINT_REDUCE = 1,/* per-point J^T J, J^T r */
    SOSLAM_STAGE_SCHUR = 2,       /* point inverse + reduced camera system */
    SOSLAM_STAGE_ALLREDUCE = 3,   /* sum of the reduced system over ranks */
    SOSLAM_STAGE_SOLVE = 4,       /* reduced camera solve (Cholesky or PCG) */
    SOSLAM_STAGE_BACKSUB = 5,     /* point back-substitution + candidate */
    SOSLAM_STAGE_COST = 6,        /* cost at the candidate */
    SOSLAM_STAGE_SYNC = 7         /* scalar read-back + host decision */
};

typedef struct soslam_ba_iteration {
    double cost;
    double candidate_cost;
    double model_cost_change;
    double relative_decrease;
    double radius;
    double step_norm;
    double gradient_max_norm;
    int32_t accepted;
    int32_t valid;
    int32_t linear_iterations;   /* PCG iterations, 0 for Cholesky */
    int32_t reserved;
} soslam_ba_iteration;

typedef struct soslam_ba_summary {
    double  initial_cost;
    double  final_cost;
    int32_t iterations;
    int32_t accepted;
    int32_t termination;
    int32_t line_search_steps;   /* Solver::Summary::num_line_search_steps: iterations of the Armijo search Ceres runs on bounded
                                    problems before a step is judged (/root/reference/src/bundle_adjuster.cpp:104-108) */
    int32_t linear_solver;       /* solver actually used */
    int32_t linear_iterations;   /* total PCG iterations */
    double  solve_seconds;       /* host wall time of the LM loop (device work included) */
    double  setup_seconds;       /* index construction + upload in set_problem */
    double  stage_ms[SOSLAM_BA_NUM_STAGES]; /* HIP-event time per stage, summed over iterations (profile_stages) */
    int32_t stage_calls[SOSLAM_BA_NUM_STAGES];
} soslam_ba_summary;

typedef struct soslam_ba soslam_ba;

const char* soslam_version(void);
const char* soslam_status_string(int status);
/* Message of the last error raised on the calling thread ("" if none). */
const char* soslam_last_error(void);

void soslam_ba_options_default(soslam_ba_options* opts);

int  soslam_ba_create(const soslam_ba_options* opts, soslam_ba** out);
void soslam_ba_destroy(soslam_ba* h);
/* Replace the options of a live handle (device and stream stay).  A handle kept across calls - one window after
 * another, /root/reference/src/slam.cpp:121-129 - re-uses its device allocations; changing linear_solver takes
 * effect at the next set_problem. */
int  soslam_ba_set_options(soslam_ba* h, const soslam_ba_options* opts);

/* Replaces ReprojectionError::SetLeftProjection / SetRightProjection
 * (/root/reference/src/reprojection_error.h:43-51): explicit problem data instead of process globals. */
int soslam_ba_set_projection(soslam_ba* h, const double* proj_l, const double* proj_r);

/*
 * Replaces the AddResidualBlock / SetParameterBlockConstant loop
 * (/root/reference/src/bundle_adjuster.cpp:62-113): uploads the observation graph and builds the
 * camera-major and point-major indices.  obs_cam[k] < n_cam, obs_pt[k] < n_pt; cam_fixed[c] != 0 holds
 * camera c constant (the reference fixes the first camera of the window; NULL = none fixed).
 * In a multi-GPU job every rank passes ALL cameras and its own shard of points and observations.
 */
int soslam_ba_set_problem(soslam_ba* h, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                          const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                          const uint8_t* cam_fixed);

/* Upload poses[n_cam*6] and points[n_pt*3]; resets the trust region to its initial radius. */
int soslam_ba_set_state(soslam_ba* h, const double* poses, const double* points);
int soslam_ba_get_state(soslam_ba* h, double* poses, double* points);

/* Replaces ceres::Solve (/root/reference/src/bundle_adjuster.cpp:116): LM until a termination test fires. */
int soslam_ba_solve(soslam_ba* h, soslam_ba_summary* summary);

/* Exactly n LM iterations from the current state, no termination tests, trust region carried over
 * (bench.py's "step").  summary may be NULL. */
int soslam_ba_iterate(soslam_ba* h, int32_t n, soslam_ba_summary* summary);

/* Per-iteration log of the last solve/iterate call; entry 0 is the initial evaluation. */
int soslam_ba_get_iteration_log(soslam_ba* h, soslam_ba_iteration* out, int32_t capacity, int32_t* count);

/*
 * One call = the body of BundleAdjuster::Optimize between gather and write-back
 * (/root/reference/src/bundle_adjuster.cpp:60-118): create, upload, solve, download, destroy.
 * poses/points are updated in place on success and untouched on failure.
 */
int soslam_ba_optimize(const soslam_ba_options* opts, const double* proj_l, const double* proj_r,
                       uint32_t n_cam, double* poses, uint32_t n_pt, double* points,
                       uint32_t n_obs, const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                       const uint8_t* cam_fixed, soslam_ba_summary* summary);

/* ---- multi-GPU: observations sharded by point, one process per GPU ---------- */

enum { SOSLAM_REDUCE_SUM = 0, SOSLAM_REDUCE_MAX = 1 };
/*
 * In-place all-reduce of count f64 values at device_buffer, enqueued on `stream` (or ordered after it).
 * The host side supplies it: RCCL's ncclAllReduce in a C++ host, torch.distributed.all_reduce on a
 * tensor aliasing the buffer in bench.py.  Return 0 on success.
 */
typedef int (*soslam_allreduce_fn)(void* user, void* device_buffer, uint64_t count, int32_t op, void* stream);
int soslam_ba_set_allreduce(soslam_ba* h, soslam_allreduce_fn fn, void* user, int32_t rank, int32_t world);
/*
 * The same exchange through HOST memory, for hosts whose collective cannot take device pointers (MPI without device
 * support; gloo in the tests): the library copies the range to a pinned buffer of its own on its own stream, calls fn
 * for an in-place all-reduce of that host range, and uploads the result.  Wins over a device callback; the RCCL leg wins
 * over both.
 */
typedef int (*soslam_host_allreduce_fn)(void* user, double* host_buffer, uint64_t count, int32_t op);
int soslam_ba_set_host_allreduce(soslam_ba* h, soslam_host_allreduce_fn fn, void* user, int32_t rank, int32_t world);
/*
 * The library's own collective leg: RCCL over xGMI (librccl.so.1 bound with dlopen at first use - no link-time
 * dependency).  One process per GPU; rank 0 draws a unique id (ncclGetUniqueId) and hands its 128 bytes to the other
 * ranks by whatever side channel the host has (MPI_Bcast, a file, a torch.distributed store); then EVERY rank calls
 * soslam_ba_init_rccl (ncclCommInitRank on the handle's device: a collective call).  From then on the two all-reduces
 * of an LM iteration are ncclAllReduce calls in place on the handle's stream; a callback set with
 * soslam_ba_set_allreduce is ignored.  The communicator is destroyed with the handle.
 */
#define SOSLAM_RCCL_UNIQUE_ID_BYTES 128
int soslam_rccl_get_unique_id(void* id128);
int soslam_ba_init_rccl(soslam_ba* h, const void* id128, int32_t rank, int32_t world);
/*
 * Sharded jobs: agree on a status word before a collective phase.  Every rank passes its own status (SOSLAM_OK or the error its
 * set-up calls returned); *agreed receives the MAX over the ranks through the attached collective leg (one 8-byte all-reduce).
 * A rank whose soslam_ba_set_problem / set_state failed must NOT simply skip soslam_ba_solve - the other ranks would wait for it
 * in their first all-reduce, and RCCL has no time-out: all ranks call this instead and leave together when *agreed != SOSLAM_OK.
 * Without a collective attached *agreed = local_status.  (A rank that cannot join the communicator at all - a failed
 * soslam_ba_init_rccl - is beyond what the library can agree on: that needs the host's launcher.)
 */
int soslam_ba_agree_status(soslam_ba* h, int local_status, int* agreed);
/*
 * After a sharded solve: poses[n_cam*6] (replicated, may be NULL) and the points of ALL ranks, points_global[n_pt_global*3],
 * on every rank - this rank's shard is the global range [shard_begin, shard_begin + n_pt).  One all-reduce (sum of
 * zero-padded shards) through the RCCL leg or the callback; with one rank it is a plain download.
 */
int soslam_ba_get_state_global(soslam_ba* h, double* poses, uint32_t n_pt_global, uint32_t shard_begin, double* points_global);
/* The per-iteration reduce payload (reduced camera system + gradient + scalars) lives in one device
 * buffer.  Query its size after set_problem; optionally make the library use a caller-owned buffer
 * (e.g. a torch tensor) of at least that many f64 so the host can alias it. */
int soslam_ba_reduce_buffer_count(soslam_ba* h, uint64_t* count_f64);
int soslam_ba_set_reduce_buffer(soslam_ba* h, void* device_ptr, uint64_t count_f64);
/*
 * Every rank must lay the reduced camera system out identically, so its block-sparsity pattern has to be
 * the job-wide one, not the shard's: pass the camera pairs (a[i], b[i]) that share a point ANYWHERE in the
 * job before soslam_ba_set_problem (the union with the shard's own pairs is used; pairs with a fixed
 * camera are ignored).  n_pairs = 0 clears it.  Single-GPU callers never need this.
 */
int soslam_ba_set_covisibility(soslam_ba* h, uint64_t n_pairs, const uint32_t* cam_a, const uint32_t* cam_b);
/* Contiguous point shard [begin, end) of rank `rank` in a `world`-rank job. */
void soslam_ba_shard_range(uint32_t n_pt, int32_t rank, int32_t world, uint32_t* begin, uint32_t* end);

/* ---- stage-level access: parity tests, roofline measurement ------------------------------------------ */

enum {
    SOSLAM_KERNEL_LINEARIZE = 0,     /* ba_linearize: reads 48 B, writes the compact row [A^T A | A^T r] = 80 B per observation */
    SOSLAM_KERNEL_COST = 1,          /* ba_cost: residual + loss only (48 B / observation) */
    SOSLAM_KERNEL_POINT_REDUCE = 2,
    SOSLAM_KERNEL_SCHUR = 3,
    SOSLAM_KERNEL_BACKSUB = 4
};
/* Launch one kernel `reps` times on the handle's stream between two HIP events recorded on that
 * stream and return the average duration in milliseconds.  State is not advanced. */
int soslam_ba_time_kernel(soslam_ba* h, int32_t kernel, int32_t reps, float* avg_ms);

enum {
    SOSLAM_DBG_RESIDUALS = 0,   /* n_obs*4  f64, caller's observation order, loss-corrected */
    SOSLAM_DBG_JAC_CAM = 1,     /* n_obs*24 f64, 4x6 row-major, zero for fixed cameras (J_c = [A D | A] from ba_linearize's own r, A, D) */
    SOSLAM_DBG_JAC_POINT = 2,   /* n_obs*12 f64, 4x3 row-major */
    SOSLAM_DBG_COST = 3,        /* 1 f64: cost at the current state */
    SOSLAM_DBG_S_DENSE = 4,     /* (6F)^2 f64 row-major, both triangles, damping included (last step) */
    SOSLAM_DBG_RHS = 5,         /* 6F f64 */
    SOSLAM_DBG_STEP_CAM = 6,    /* n_cam*6 f64 (zero rows for fixed cameras) */
    SOSLAM_DBG_STEP_POINT = 7,  /* n_pt*3 f64, caller's point order */
    SOSLAM_DBG_STEP_SCALARS = 8,/* 6 f64: cost, model_cost_change, candidate_cost, step_norm, linear-solver
                                   iterations, linear-solver status (0 ok) */
    SOSLAM_DBG_COMPACT_ROWS = 9 /* n_obs*9 f64, caller's observation order: the compact row [Gxx Gxy Gxz Gyy Gyz Gzz | hx hy hz]
                                   (G = A^T A, h = A^T r) of every observation AS STORED by ba_linearize - the array every
                                   later kernel of the iteration reads */
};
/* Evaluate at the current state without advancing it: linearise, and for the S/RHS/STEP items take one
 * trust-region step with `radius` and scaling from this linearisation (mirrors oracle_ba_step). */
int soslam_ba_debug_step(soslam_ba* h, double radius);
int soslam_ba_debug_read(soslam_ba* h, int32_t what, void* dst, uint64_t bytes);

/* ---- host-side conversions of the reference's math_utils.h, float32 arithmetic ------------------------ */

/* camera->world Matrix4f (row-major) -> world->camera pose[6]: GlobalPose().inverse() then MatrixToPose
 * (/root/reference/src/bundle_adjuster.cpp:66-68, /root/reference/src/math_utils.h:12-25). */
void soslam_pose_from_global_matrix(const float* t_wc16, double* pose6);
/* pose[6] -> camera->world Matrix4f: PoseToMatrix then inverse
 * (/root/reference/src/math_utils.h:27-41, /root/reference/src/bundle_adjuster.cpp:122-125). */
void soslam_global_matrix_from_pose(const double* pose6, float* t_wc16);

#ifdef __cplusplus
}
#endif
#endif
