// linsolve.h - reduced camera system solvers on the device (definitions in linsolve.hip).
//
// The reference solves this system with SPARSE_SCHUR + CHOLMOD (/root/reference/src/bundle_adjuster.cpp:26-29),
// a direct factorisation.  Two device paths replace it:
//   dense Cholesky  - block-sparse S expanded to a dense lower triangle, blocked right-looking
//                     factorisation (BASELINE.json configs[1]: "full Schur complement + dense camera solve");
//   PCG             - block-Jacobi preconditioned conjugate gradients on the block-sparse S
//                     (configs[2]: "PCG on Schur-complemented system").
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace soslam {

// Symmetric block-sparse matrix, 6x6 row-major blocks stored once for the upper triangle (i <= j).
// Row lists hold BOTH triangles: entry e of block-row i refers to block ent_blk[e] in column ent_col[e],
// used transposed when ent_trans[e] != 0.
struct BsrView {
    uint32_t n_rows;             // block rows (free cameras)
    const uint32_t* row_ptr;     // [n_rows + 1]
    const uint32_t* ent_col;     // [n_ent]
    const uint32_t* ent_blk;     // [n_ent]
    const uint8_t* ent_trans;    // [n_ent]
    const int32_t* diag_block;   // [n_rows]
    const double* blocks;        // [n_blocks][36]
};

// x = S^-1 b by block-Jacobi PCG; resid receives the final recurrence residual b - S x.
// work: 4 * 6 * n_rows f64 (p, z, q, spare) + 36 * n_rows (preconditioner).
// scal[SC_LIN_ITERS], scal[SC_LIN_RESID] (relative), scal[SC_LIN_STATUS] are written.
void launch_pcg(hipStream_t s, const BsrView& A, const double* b, double* x, double* resid, double* work,
                double tol, int max_iter, double* scal);

// Dense path: expand, factor (lower Cholesky, in place), solve.  dense is (6 n_rows)^2 f64.
// scal[SC_LIN_STATUS] = 1 on a non-positive pivot.
void launch_bsr_to_dense(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row,
                         const uint32_t* blk_col, double* dense);
void launch_dense_cholesky_solve(hipStream_t s, uint32_t n, double* dense, const double* b, double* x, double* scal);

// up to 1 260 unknowns: S padded to tiles of 60, inverted explicitly (block Gauss-Jordan on the matrix cores), two dense products
// and one refinement against the block-sparse S; work: dense_inverse_count(n_rows) f64
bool dense_inverse_fits(uint32_t n_rows);
size_t dense_inverse_count(uint32_t n_rows);
void launch_dense_inverse_solve(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col,
                                double* work, const double* b, double* x, double* scal);

size_t pcg_work_count(uint32_t n_rows);

// Small systems (<= kDenseSmallRowsMax block rows: the reference's sliding windows): S expanded, factored and solved by
// ONE workgroup in LDS, one launch.  scal[SC_LIN_STATUS] = 1 on a non-positive pivot.
constexpr int kDenseSmallRowsMax = 22;   // 132 unknowns: 132 x 133 f64 = 140 KB of the 160 KB LDS
bool dense_small_fits(uint32_t n_rows);
// <= 20 block rows: two-block elimination on the matrix cores by one workgroup (crsolve.hip), same interface
bool dense2_fits(uint32_t n_rows);
// dense symmetric positive definite inverse in place by block Gauss-Jordan over 60 x 60 tiles (crsolve.hip): A is (60 nb)^2
// row-major, ebuf 2 x 3600 f64 of scratch, status[0] = 1 when a pivot tile was not positive definite
void launch_dense_spd_inverse60(hipStream_t s, double* A, int nb, double* ebuf, double* status);
void launch_dense2_solve(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col,
                         const double* b, double* x, double* scal);
void launch_dense_small_solve(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col,
                              const double* b, double* x, double* scal);

}  // namespace soslam

namespace soslam {

// ---- block-band path ---------------------------------------------------------------------------------
// Camera chains (visual odometry windows) give a reduced matrix whose blocks lie within `bw` block
// diagonals.  It is factorised directly (like the reference's CHOLMOD solve) by ONE workgroup that slides
// an LDS window of (bw+2) block rows down the band; the factor also serves PCG as its preconditioner.
// Storage: band[(i*(bw+1) + d)*36 + a*6 + b] = S(6i+a, 6(i-d)+b), d = 0..bw (lower block band).
constexpr int kBandMax = 15;   // widest block half-bandwidth the LDS window admits (78 KB at 15)

size_t band_count(uint32_t n_rows, int bw);                 // f64 in the band array (and in bandT)
size_t pcg_band_work_count(uint32_t n_rows);                // f64 of PCG work space
void launch_bsr_to_band(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row,
                        const uint32_t* blk_col, int bw, double* band);
// in-place Cholesky of the band.  bandT receives the factor column-oriented (bandT[k][d] = L(k+d, k)) for the
// forward solve, dinv[n_rows][36] the inverses of the diagonal factor blocks.
// scal[SC_LIN_STATUS] = 1 on a non-positive pivot.
void launch_band_cholesky(hipStream_t s, uint32_t n_rows, int bw, double* band, double* bandT, double* dinv, double* scal);
// x = (L L^T)^-1 b with the factor above; one wave
void launch_band_solve(hipStream_t s, uint32_t n_rows, int bw, const double* band, const double* bandT, const double* dinv,
                       const double* b, double* x);
// PCG on the block-sparse S preconditioned by the band factor: up to max_rounds rounds of
// {preconditioner solve (one wave), S p over many workgroups, update + convergence test}; a round whose
// predecessor converged returns at once on a device-side flag, so the host never synchronises.  With every
// block of S inside the band the first round already reaches rounding level.
void launch_pcg_band(hipStream_t s, const BsrView& A, int bw, const double* band, const double* bandT, const double* dinv,
                     const double* b, double* x, double* resid, double* work, double tol, int max_rounds, double* scal);

// ---- block cyclic reduction (crsolve.hip): the same band, factored level by level over many workgroups ---------
// Usable for bw <= kCrBandMax; the sequential band kernels above remain the path for wider bands.
constexpr int kCrBandMax = 10;
size_t cr_count(uint32_t n_rows, int bw);   // f64 of workspace (D, F, E, P, Q, P^T, Q^T super-blocks + padded vector)
size_t cr_map_count(uint32_t n_rows, int bw);   // int32 entries of the gather map
void cr_build_map(uint32_t n_rows, int bw, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col, int32_t* map);   // host
// damp != NULL: the gather also applies the camera damping to S in place (instead of a launch_cam_damp before it)
struct CamDamp;
// fwd_b != NULL: the factorisation also carries the forward sweep of the solve that follows for this right-hand side (the
// levels launch_cr_solve would run as pairs); that solve must then be called with forward_done = true and the same b
// comp != NULL and comp_scale > 0 (off-band mode: some blocks of A lie outside the factored band): comp_scale * comp is added to
// the diagonal of the factored matrix
void launch_cr_factor(hipStream_t s, const BsrView& A, const int32_t* map, int bw, double* ws, double* scal, const CamDamp* damp,
                      const double* fwd_b = nullptr, const double* comp = nullptr, double comp_scale = 0.0);
// off-band mode: comp = absolute row sums of the blocks the map leaves out (lists from cr_build_comp_lists)
void cr_build_comp_lists(uint32_t n_rows, int bw, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col, std::vector<uint32_t>& ptr,
                         std::vector<uint32_t>& ent);
void launch_cr_comp(hipStream_t s, const double* blocks, uint32_t n_rows, const uint32_t* ptr, const uint32_t* ent, double* comp);
// x = S^-1 b with the factors in ws; every launch returns at once when *done_flag != 0 (may be NULL)
void launch_cr_solve(hipStream_t s, uint32_t n_rows, int bw, double* ws, const double* b, double* x, const double* done_flag,
                     bool forward_done = false);
// PCG rounds as launch_pcg_band, preconditioner = the cyclic-reduction factor
void launch_pcg_cr(hipStream_t s, const BsrView& A, int bw, double* cr_ws, const double* b, double* x, double* resid, double* work,
                   double tol, int max_rounds, double* scal, bool forward_done = false);
void launch_pcg_cr_more(hipStream_t s, const BsrView& A, int bw, double* cr_ws, double* x, double* resid, double* work, double tol, int rounds,
                        double* scal);

}  // namespace soslam

namespace soslam {

// ---- multi-workgroup block-Jacobi PCG on (A + shift I) ---------------------------------------------------
// For systems too large or too irregular for the band factor (pose graphs with loop closures, BA with a wide
// co-visibility pattern).  Every vector operation and the product with A run over many workgroups; the two
// reductions of an iteration are per-workgroup partials summed redundantly (fixed order) by every workgroup
// of the next kernel, so there is no single-CU stage and no host round trip inside a chunk of iterations.
size_t pcg_multi_work_count(uint32_t n_rows);   // f64 of caller-provided work space
// Runs until |r| <= tol |b| or max_iter; synchronises the stream once per `chunk` iterations to read the
// device-side state.  Returns iterations done (>= 0) or -1 on breakdown; *rel_resid receives |r|/|b|.
// resid (required: it is the iteration's residual vector) receives the final b - (A + shift I) x.
int pcg_multi_solve(hipStream_t s, const BsrView& A, double shift, const double* b, double* x, double* resid,
                    double* work, double tol, int max_iter, int chunk, double* rel_resid);

}  // namespace soslam

namespace soslam {

// ---- two-level PCG (pcg_multi.hip): block-Jacobi + an aggregation coarse space with an explicit coarse inverse ---------------
struct TwoLevelView {
    uint32_t n_agg;              // aggregates (one workgroup each; at most 42 block rows, contiguous)
    uint32_t ncp;                // coarse dimension padded to a multiple of 60 (launch_dense_spd_inverse60), <= 1260
    const uint32_t* agg_ptr;     // [n_agg + 1] first block row of every aggregate
    const double* P;             // [n_rows][36] row-major: the block row's increment per coarse unknown of its aggregate
    const double* Ainv;          // [ncp * ncp] the coarse operator (pcg2_coarse_inverse)
    double* rc;                  // [ncp] P^T r (entries from 6 n_agg on stay zero)
};
size_t pcg2_work_count(uint32_t n_rows, uint32_t n_agg);
// set-up of the coarse space: lists (host, once per pattern) and G = P^T P, Ac0 = P^T A P (device, once per matrix)
void two_level_lists(uint32_t n_agg, const uint32_t* row_agg, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col,
                     std::vector<uint32_t>& cb_ptr, std::vector<uint32_t>& cb_ent, std::vector<uint32_t>& cb_I, std::vector<uint32_t>& cb_J);
void launch_coarse_assemble(hipStream_t s, uint32_t n_agg, const uint32_t* agg_ptr, uint32_t n_cb, const uint32_t* cb_ptr, const uint32_t* cb_ent,
                            const uint32_t* cb_I, const uint32_t* cb_J, const uint32_t* blk_row, const uint32_t* blk_col, const double* blocks,
                            const double* P, uint32_t ncp, double* G, double* Ac0);
// the coarse operator (P^T A P + shift P^T P)^-1 into ainv_out: Ac0 = P^T A P ([ncp * ncp], without the shift), G = P^T P per
// aggregate ([n_agg][36]), ebuf 2 x 3600 f64 of scratch, status[0] = 1 on a non-positive pivot tile.  Any symmetric positive
// definite coarse operator makes a valid preconditioner: the caller may hand pcg2_solve the inverse of an EARLIER matrix.
void pcg2_coarse_inverse(hipStream_t s, uint32_t n_agg, uint32_t ncp, const double* Ac0, const double* G, double shift, double* ainv_out,
                         double* ebuf, double* status);
// as pcg_multi_solve, preconditioner D^-1 + P tl.Ainv P^T
int pcg2_solve(hipStream_t s, const BsrView& A, double shift, const double* b, double* x, double* resid, double* work, const TwoLevelView& tl,
               double tol, int max_iter, int chunk, double* rel_resid);

}  // namespace soslam
