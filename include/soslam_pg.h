/*
 * soslam_pg.h - C ABI of the MI355X-native pose-graph backend.
 *
 * Replaces the g2o solve inside PoseGraphOptimizer::Optimize()
 * (/root/reference/src/pose_graph_optimizer.cpp:61-69: gauge fix, initializeOptimization(), optimize(10)) and
 * the solver the constructor configures (:6-27: OptimizationAlgorithmLevenberg + BlockSolverX + LinearSolverEigen,
 * information diag(.01,.01,.01,1,1,1), one shared RobustKernelHuber).  Graph construction stays on the host
 * (stereo_orb_slam_amd/host/pose_graph_optimizer.cpp keeps the class API); loop-edge MEASUREMENTS come in as data
 * because the reference derives them with its image front-end (:175-249), which is out of scope.
 *
 * Conventions (g2o VertexSE3 / EdgeSE3):
 *   estimate / measurement  double[7] = tx ty tz qx qy qz qw  (/root/reference/src/pose_graph_optimizer.cpp:108-116)
 *   edge k                  from[k] -> to[k]: error = toVectorMQT(Z_k^-1 * X_from^-1 * X_to)
 *   information             one row-major 6x6 shared by all edges, translation rows first
 *   fixed[v] != 0           vertex v is held constant (the reference fixes vertex 0, :118-121)
 * Plain pointers and sizes; arrays are caller-allocated and updated only on success.
 */
#ifndef SOSLAM_PG_H
#define SOSLAM_PG_H

#include <stdint.h>

#include "soslam_ba.h" /* status codes, soslam_last_error */

#ifdef __cplusplus
extern "C" {
#endif

enum { SOSLAM_PG_TERM_ITERATIONS = 0, SOSLAM_PG_TERM_TRIALS = 1, SOSLAM_PG_TERM_FAILURE = 2 };
/* preconditioner of the PCG that stands in for g2o's LinearSolverEigen (a direct sparse Cholesky,
 * /root/reference/src/pose_graph_optimizer.cpp:14-18): block-Jacobi; block-Jacobi plus a coarse space of six rigid-body modes
 * per aggregate of neighbouring vertices (two-level); or - a chain of keyframes with loop closures, the reference's own graphs -
 * the exact factor of H as a band: numbered breadth-first such a graph has a half-bandwidth of a few vertices whatever the length
 * of its loops (block cyclic reduction; edges the band of ten leaves out stay in the matrix-vector product).  AUTO: the band factor
 * when at most four edges (and at most 1 %) lie outside the band, else two-level from 64 free vertices on, else block-Jacobi */
enum { SOSLAM_PG_PRECOND_AUTO = 0, SOSLAM_PG_PRECOND_BLOCK_JACOBI = 1, SOSLAM_PG_PRECOND_TWO_LEVEL = 2, SOSLAM_PG_PRECOND_BAND_FACTOR = 3 };

typedef struct soslam_pg_options {
    int32_t max_iterations;      /* 10: optimize(10), /root/reference/src/pose_graph_optimizer.cpp:69 */
    int32_t max_trials;          /* 10: g2o maxTrialsAfterFailure */
    double  huber_delta;         /* 1.0: RobustKernelHuber default */
    double  tau;                 /* 1e-5: lambda0 = tau * max diag(H) */
    double  pcg_tolerance;       /* 1e-10: relative residual of the linear solve (the reference solves directly) */
    int32_t pcg_max_iterations;  /* 4000 */
    int32_t verbose;             /* 1 = one line per iteration (setVerbose(true), :21) */
    int32_t device;              /* -1 = current */
    int32_t preconditioner;      /* SOSLAM_PG_PRECOND_*; 0 = AUTO */
    void*   stream;              /* hipStream_t; NULL = own stream */
} soslam_pg_options;

typedef struct soslam_pg_iteration {
    double  chi2;               /* robust chi2 after the iteration */
    double  lambda;
    int32_t trials;
    int32_t accepted;
    int32_t linear_iterations;
    int32_t reserved;
} soslam_pg_iteration;

typedef struct soslam_pg_summary {
    double  initial_chi2;
    double  final_chi2;
    int32_t iterations;
    int32_t termination;
    int32_t linear_iterations;
    int32_t reserved;
    double  solve_seconds;
    double  setup_seconds;
    double  linearize_ms;       /* HIP-event time of the edge linearisation kernel, summed */
    double  linear_solve_ms;    /* HIP-event time of the PCG launches, summed */
} soslam_pg_summary;

typedef struct soslam_pg soslam_pg;

void soslam_pg_options_default(soslam_pg_options* opts);
int  soslam_pg_create(const soslam_pg_options* opts, soslam_pg** out);
void soslam_pg_destroy(soslam_pg* h);

/* Upload the whole graph (the g2o optimizer of the reference persists and grows between calls; the host shim
 * keeps the accumulated vertices/edges and uploads them again - a few hundred kB). */
int soslam_pg_set_graph(soslam_pg* h, uint32_t n_vertex, const double* est, const uint8_t* fixed, uint32_t n_edge,
                        const uint32_t* e_from, const uint32_t* e_to, const double* meas, const double* info36);
/*
 * The reference's optimizer persists and GROWS between calls (m_last_id, /root/reference/src/pose_graph_optimizer.cpp:56-59):
 * append new vertices (with their initial estimates and fixed flags) and new edges to the handle's graph.  Existing
 * vertices keep the estimates the last soslam_pg_optimize left on the device, as g2o's vertices do.  The first call
 * (empty handle) needs info36; later calls may pass NULL to keep it.  Edge endpoints index the grown vertex list.
 */
int soslam_pg_append(soslam_pg* h, uint32_t n_add_vertex, const double* est_add, const uint8_t* fixed_add, uint32_t n_add_edge,
                     const uint32_t* e_from, const uint32_t* e_to, const double* meas, const double* info36);
int soslam_pg_graph_size(soslam_pg* h, uint32_t* n_vertex, uint32_t* n_edge);
/* g2o's optimize(max_iterations): Levenberg iterations on the uploaded graph */
int soslam_pg_optimize(soslam_pg* h, soslam_pg_summary* summary);
int soslam_pg_get_estimates(soslam_pg* h, double* est);
int soslam_pg_get_iteration_log(soslam_pg* h, soslam_pg_iteration* out, int32_t capacity, int32_t* count);

/* create + set_graph + optimize + get_estimates + destroy; est is updated in place on success only */
int soslam_pg_solve(const soslam_pg_options* opts, uint32_t n_vertex, double* est, const uint8_t* fixed, uint32_t n_edge,
                    const uint32_t* e_from, const uint32_t* e_to, const double* meas, const double* info36,
                    soslam_pg_summary* summary);

/* parity tests: per-edge error (n_edge*6) and Jacobians (n_edge*36 each, row-major) at the current estimates,
 * and the assembled system: robust chi2, dense H ((6F)^2, F = free vertices, no damping), b (6F). Any pointer may
 * be NULL. */
int soslam_pg_debug_linearize(soslam_pg* h, double* edge_e, double* edge_ji, double* edge_jj, double* chi2,
                              double* h_dense, double* b);
/* average duration (ms) of `reps` launches of the edge linearisation kernel, HIP events on the handle's stream */
int soslam_pg_time_linearize(soslam_pg* h, int32_t reps, float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif
