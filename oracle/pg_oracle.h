/*
 * pg_oracle.h - CPU oracle for the pose-graph path (PoseGraphOptimizer::Optimize's g2o solve).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED: the arithmetic lives in g2o (unvendored, taken from
 * ../../libs/g2o outside the reference tree, /root/reference/CMakeLists.txt:73,108-111; call sites
 * /root/reference/src/pose_graph_optimizer.cpp:14-21,61-69,113-124,140-171).  Semantics restated from
 * VertexSE3 / EdgeSE3 (error = toVectorMQT(Z^-1 Xi^-1 Xj), update X <- X * fromVectorMQT),
 * RobustKernelHuber on chi2 = e^T Omega e, OptimizationAlgorithmLevenberg (lambda0 = 1e-5 max diag H, up to 10
 * trials per iteration, rho = (chi - chi') / (x.(lambda x + b) + 1e-3)).  Pinned only by the torch-autograd
 * golden vectors of oracle/gen_golden.py (tests/golden/pg_edge_jacobian.npz) and by scipy on small graphs.
 */
#ifndef PG_ORACLE_H
#define PG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_pg_options {
    int32_t max_iterations;       /* optimize(10), /root/reference/src/pose_graph_optimizer.cpp:69 */
    int32_t max_trials;           /* g2o maxTrialsAfterFailure = 10 */
    double  huber_delta;          /* RobustKernelHuber default delta = 1 (shared kernel, :167,209) */
    double  initial_lambda_scale; /* g2o tau = 1e-5 */
    int32_t num_threads;
    int32_t reserved;
} oracle_pg_options;

typedef struct oracle_pg_iteration {
    double  chi2;      /* robust chi2 after the iteration */
    double  lambda;    /* lambda after the iteration */
    int32_t trials;    /* levenbergIterations of this iteration */
    int32_t accepted;
} oracle_pg_iteration;

enum { ORACLE_PG_TERM_ITERATIONS = 0, ORACLE_PG_TERM_TRIALS = 1, ORACLE_PG_TERM_FAILURE = 2 };

typedef struct oracle_pg_summary {
    double  initial_chi2;
    double  final_chi2;
    int32_t iterations;
    int32_t termination;
    double  solve_seconds;
    double  setup_seconds;
} oracle_pg_summary;

void oracle_pg_options_default(oracle_pg_options* o);

/* error e[6] and the two 6x6 Jacobians (row-major, wrt the [dt, dq.xyz] increments of xi and xj) of one edge;
 * xi, xj, z are [t(3), q.xyz, q.w]. */
void oracle_pg_edge(const double* xi7, const double* xj7, const double* z7, double* e6, double* ji36, double* jj36);

/* sum over edges of rho(e^T Omega e); per-edge chi2 (non-robust) optionally written. info is one shared 6x6. */
double oracle_pg_chi2(uint32_t n_vertex, uint32_t n_edge, const double* est, const uint32_t* e_from, const uint32_t* e_to,
                      const double* meas, const double* info36, double huber_delta, double* edge_chi2);

/* dense system at the current estimates: H ((6F)^2, F = free vertices, no damping) and b (6F); returns robust chi2 */
double oracle_pg_linearize(uint32_t n_vertex, uint32_t n_edge, const double* est, const uint8_t* fixed,
                           const uint32_t* e_from, const uint32_t* e_to, const double* meas, const double* info36,
                           double huber_delta, double* h_dense, double* b);

/* g2o's optimize(max_iterations) with OptimizationAlgorithmLevenberg; est is updated in place */
int oracle_pg_solve(uint32_t n_vertex, uint32_t n_edge, double* est, const uint8_t* fixed, const uint32_t* e_from,
                    const uint32_t* e_to, const double* meas, const double* info36, const oracle_pg_options* opt,
                    oracle_pg_summary* summary, oracle_pg_iteration* iter_log);

#ifdef __cplusplus
}
#endif
#endif
