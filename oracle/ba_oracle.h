/*
 * ba_oracle.h - CPU oracle for the bundle-adjustment hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped HIP path
 * never calls into it.
 *
 * PARITY UNPINNED: the arithmetic this restates lives in Ceres Solver
 * (unvendored, version unpinned, >= 2.0 inferred; call sites
 * /root/reference/src/bundle_adjuster.cpp:60,100,102,106-107,113,116-118 and
 * /root/reference/src/reprojection_error.h:20,58).  The reference holds no
 * test, fixture or golden vector for this path (SURVEY.md section 0 fact 5,
 * section 8(c)) and cannot be built here, so the oracle is pinned only by
 * independent restatements made in this repository (oracle/gen_golden.py:
 * torch-f64 autograd of the residual formula, scipy least_squares minima,
 * numpy dense normal equations), committed under tests/golden/.
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ba_options {
    int32_t max_iterations;        /* BA_MAX_ITERATION = 50, /root/reference/src/params.h:34 */
    int32_t check_termination;     /* 0: run exactly max_iterations LM iterations (timing runs) */
    double  huber_delta;           /* HuberLoss(1.0), /root/reference/src/bundle_adjuster.cpp:100 */
    double  lower_bound;           /* -1e4, /root/reference/src/params.h:44 */
    double  upper_bound;           /* +1e4, /root/reference/src/params.h:47 */
    double  initial_radius;        /* Ceres default 1e4 */
    double  max_radius;            /* 1e16 */
    double  min_radius;            /* 1e-32 */
    double  min_relative_decrease; /* 1e-3 */
    double  min_lm_diagonal;       /* 1e-6 */
    double  max_lm_diagonal;       /* 1e32 */
    double  parameter_tolerance;   /* 1e-8 */
    double  function_tolerance;    /* 1e-16, /root/reference/src/bundle_adjuster.cpp:36 */
    double  gradient_tolerance;    /* 1e-16, /root/reference/src/bundle_adjuster.cpp:35 */
    int32_t jacobi_scaling;        /* 1 */
    int32_t num_threads;           /* OpenMP threads for the per-observation loops; 1 = scalar */
} oracle_ba_options;

enum {
    ORACLE_TERM_MAX_ITERATIONS = 0,
    ORACLE_TERM_PARAMETER_TOLERANCE = 1,
    ORACLE_TERM_FUNCTION_TOLERANCE = 2,
    ORACLE_TERM_GRADIENT_TOLERANCE = 3,
    ORACLE_TERM_MIN_RADIUS = 4,
    ORACLE_TERM_INVALID_STEPS = 5,
    ORACLE_TERM_FAILURE = 6
};

typedef struct oracle_ba_iteration {
    double cost;            /* cost of the current iterate after this iteration */
    double candidate_cost;
    double model_cost_change;
    double relative_decrease;
    double radius;          /* radius used for this iteration's step */
    double step_norm;
    double gradient_max_norm;
    int32_t accepted;
    int32_t valid;
} oracle_ba_iteration;

typedef struct oracle_ba_summary {
    double initial_cost;
    double final_cost;
    int32_t iterations;          /* LM iterations executed (accepted + rejected + invalid) */
    int32_t accepted;
    int32_t termination;
    int32_t line_search_steps;   /* Solver::Summary::num_line_search_steps: iterations of the bounded-problem Armijo search */
    double  solve_seconds;       /* wall time of the LM loop only */
    double  setup_seconds;       /* index construction */
} oracle_ba_summary;

void oracle_ba_options_default(oracle_ba_options* o);

/* Residual of /root/reference/src/reprojection_error.h:12-41. */
void oracle_ba_residual(const double* cam6, const double* pt3, const double* uv4,
                        const double* proj_l, const double* proj_r, double* r4);

/* Residual + exact derivatives: what AutoDiffCostFunction<ReprojectionError,4,6,3>
 * (/root/reference/src/reprojection_error.h:58) returns.  jc is 4x6 row-major, jp 4x3 row-major. */
void oracle_ba_residual_jacobian(const double* cam6, const double* pt3, const double* uv4,
                                 const double* proj_l, const double* proj_r,
                                 double* r4, double* jc24, double* jp12);

/* HuberLoss(delta) on s = |r|^2: rho[0..2] = rho, rho', rho''. */
void oracle_huber(double s, double delta, double* rho3);

/* Corrected residuals/Jacobians of every observation at (cams, pts); returns cost = 1/2 sum rho. */
double oracle_ba_linearize(uint32_t n_obs, const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                           const double* cams, const double* pts, const double* proj_l, const double* proj_r,
                           const uint8_t* cam_fixed, double huber_delta,
                           double* r_out /* n_obs*4 or NULL */, double* jc_out /* n_obs*24 or NULL */,
                           double* jp_out /* n_obs*12 or NULL */);

/* Cost only. */
double oracle_ba_cost(uint32_t n_obs, const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                      const double* cams, const double* pts, const double* proj_l, const double* proj_r,
                      double huber_delta);

/*
 * One trust-region step from (cams, pts) with the given radius and with Jacobi
 * scaling taken from this same linearisation (as on Ceres' first iteration).
 * Outputs (any may be NULL): dense reduced system s_dense (6F x 6F, row-major,
 * F = number of free cameras, both triangles, damping included), rhs (6F),
 * dc (n_cam*6, zero rows for fixed cameras), dp (n_pt*3), scalars[4] =
 * {cost, model_cost_change, candidate_cost, step_norm}.
 */
int oracle_ba_step(uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                   const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                   const double* cams, const double* pts, const double* proj_l, const double* proj_r,
                   const uint8_t* cam_fixed, const oracle_ba_options* opt, double radius,
                   double* s_dense, double* rhs, double* dc, double* dp, double* scalars);

/*
 * The whole solve: what ceres::Solve does for the problem BundleAdjuster::Optimize
 * builds (/root/reference/src/bundle_adjuster.cpp:39-118).  cams/pts are updated in
 * place.  iter_log may be NULL or hold max_iterations+1 entries (entry 0 = initial
 * evaluation).
 */
int oracle_ba_solve(uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                    const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                    double* cams, double* pts, const double* proj_l, const double* proj_r,
                    const uint8_t* cam_fixed, const oracle_ba_options* opt,
                    oracle_ba_summary* summary, oracle_ba_iteration* iter_log);

/*
 * Virtual-rank check of the multi-GPU decomposition:
 * partition the points over `n_rank` shards (point p -> shard p*n_rank/n_pt),
 * accumulate each shard's reduced-system contribution separately, sum them in
 * rank order (the all-reduce), then add the camera damping.  Same outputs as
 * oracle_ba_step's s_dense / rhs, so the two must agree to rounding.
 */
int oracle_ba_step_sharded(uint32_t n_rank, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                           const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                           const double* cams, const double* pts, const double* proj_l, const double* proj_r,
                           const uint8_t* cam_fixed, const oracle_ba_options* opt, double radius,
                           double* s_dense, double* rhs);

/*
 * What ONE rank of a point-sharded job contributes before the all-reduce: the dense reduced-system partial
 * (no camera damping: that is added after the reduction from the reduced diagonal), its right-hand side and
 * the diagonal of its camera blocks.  Summing the outputs over rank = 0..n_rank-1 and adding the camera
 * damping gives oracle_ba_step's s_dense / rhs.
 */
int oracle_ba_shard_system(uint32_t rank, uint32_t n_rank, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                           const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                           const double* cams, const double* pts, const double* proj_l, const double* proj_r,
                           const uint8_t* cam_fixed, const oracle_ba_options* opt, double radius,
                           double* s_partial, double* rhs_partial, double* diag_partial);

#ifdef __cplusplus
}
#endif
#endif
