// pg_kernels.h - launch wrappers of the pose-graph kernels (definitions in pg_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace soslam {

struct PgInfo { double m[36]; };   // shared information matrix, row-major

// where edge k scatters its blocks: free indices of its vertices (-1 = fixed), block ids in the upper
// block-sparse H, and whether the stored off-diagonal block is H(j,i) (fj < fi) or H(i,j)
struct PgEdgeBlocks {
    int32_t fi, fj, diag_i, diag_j, off, off_is_ji;
};

// per-edge record of the linearisation: [J_i^T W J_i (36) | J_j^T W J_j (36) | off-diagonal block as stored (36) | -J_i^T W e (6) |
// -J_j^T W e (6)]; parts that belong to a fixed vertex are not written (and never read)
constexpr int kPgEdgeRec = 120;
void launch_pg_linearize(hipStream_t s, uint32_t n_edge, const double* est, const uint32_t* ef, const uint32_t* et,
                         const double* meas, const PgInfo& info, double delta, const PgEdgeBlocks* eb, double* econ,
                         double* chi_part, double* dbg_e, double* dbg_ji, double* dbg_jj);
// H (every block) and b (every free vertex) = sums of the edge records through host-built lists, fixed order
void launch_pg_gather(hipStream_t s, uint32_t n_blocks, const uint32_t* ptr, const uint32_t* ent, const uint32_t* blk_row,
                      const uint32_t* blk_col, const double* econ, double* H, double* b);
void launch_pg_chi2(hipStream_t s, uint32_t n_edge, const double* est, const uint32_t* ef, const uint32_t* et,
                    const double* meas, const PgInfo& info, double delta, double* chi_part);
// out[0] = sum(part[0..n)); if diag_block: out[1] = max |diag(H)|
// H += delta I on the free vertices' diagonal blocks (the band-factor path keeps the Levenberg shift inside H: the factorisation
// and the PCG's matrix-vector product both read the shifted matrix)
void launch_pg_shift_diag(hipStream_t s, double* H, const int32_t* diag_block, uint32_t n_free, double delta);
void launch_pg_reduce(hipStream_t s, const double* part, uint32_t n, const double* H, const int32_t* diag_block, uint32_t n_free,
                      double* out);
void launch_pg_update(hipStream_t s, uint32_t n_vertex, const double* est, const int32_t* free_idx, const double* x,
                      const double* b, double lambda, double* cand, double* scale_part);

// two-level preconditioner (pcg_multi.hip): the coarse basis P of the free vertices at the current estimates
void launch_pg_coarse_basis(hipStream_t s, uint32_t n_free, const uint32_t* free_vertex, const uint32_t* row_agg, const uint32_t* agg_ref,
                            const double* est, double* P);

}  // namespace soslam
