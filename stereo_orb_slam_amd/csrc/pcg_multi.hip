// pcg_multi.hip - multi-workgroup block-Jacobi PCG on a block-sparse symmetric matrix (6x6 blocks) + shift.
//
// Two kernels per iteration, each over ceil(n_rows / 42) workgroups of 256 lanes (42 block rows = 252 scalar
// rows per workgroup):
//   matvec     convergence test and beta from the partials; p = z + beta p_old formed on the fly for every column the
//              workgroup touches and stored for its own rows (p is double-buffered by iteration parity, the old one is
//              still being read elsewhere); q = (A + shift I) p, partial p.q
//   update     alpha from the partials; x += alpha p; r -= alpha q; z = M^-1 r; partial r.z and r.r
// (the direction step was a launch of its own; a dependent launch costs about 5 us here however little it does)
// Partials are double-buffered by iteration parity; every workgroup sums them in the same fixed order, so all
// agree bit for bit on alpha, beta and on convergence, and a converged solve turns the remaining launches of a
// chunk into immediate returns.
#include "linsolve.h"
#include "reduce.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace soslam {

namespace {

constexpr int kRowsPerWg = 42;   // block rows per workgroup
constexpr int kThreads = 256;
enum { ST_BB = 0, ST_ITERS = 1, ST_BREAKDOWN = 2, ST_DONE = 3, ST_COUNT = 4 };

struct PcgBufs {
    double *p[2], *z, *q, *minv, *part_pq, *part_rz, *part_rr, *state;   // p, part_rz, part_rr: by iteration parity
    uint32_t n_wg;
};

__device__ __forceinline__ double sum_partials(const double* __restrict__ part, uint32_t n)
{
    double s = 0.0;
    for (uint32_t i = 0; i < n; i++) s += part[i];
    return s;
}

__device__ __forceinline__ double block_sum256(double v, double* red)
{
    v = wave_sum(v);
    __syncthreads();
    if (threadIdx.x % 64 == 0) red[threadIdx.x / 64] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ bool spd6_inverse_shift(const double* __restrict__ A, double shift, double* __restrict__ out)
{
    double L[6][6], Li[6][6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
#pragma unroll
        for (int i = j; i < 6; i++) {
            double s = A[i * 6 + j] + (i == j ? shift : 0.0);
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) { ok = ok && (s > 0.0); L[j][j] = sqrt(s); } else L[i][j] = s / L[j][j];
        }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        Li[j][j] = 1.0 / L[j][j];
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; k++) s -= L[i][k] * Li[k][j];
            Li[i][j] = s / L[i][i];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) {
            double s = 0.0;
#pragma unroll
            for (int k = b; k < 6; k++) s += Li[k][a] * Li[k][b];
            out[a * 6 + b] = s;
            out[b * 6 + a] = s;
        }
    return ok;
}

__device__ __forceinline__ double bsr_row_dot(const BsrView& A, uint32_t row, const double* __restrict__ v)
{
    const uint32_t f = row / 6, a = row % 6;
    double s = 0.0;
    for (uint32_t e = A.row_ptr[f]; e < A.row_ptr[f + 1]; e++) {
        const double* B = A.blocks + 36 * (size_t)A.ent_blk[e];
        const double* x = v + 6 * (size_t)A.ent_col[e];
        if (A.ent_trans[e]) {
#pragma unroll
            for (int c = 0; c < 6; c++) s += B[c * 6 + a] * x[c];
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) s += B[a * 6 + c] * x[c];
        }
    }
    return s;
}

// x = 0, r = b, M = blockdiag(A + shift I)^-1, z = M r, p = z; partials of b.b (-> rr parity 0) and r.z (parity 0)
__global__ __launch_bounds__(kThreads) void pcgm_init_kernel(const BsrView A, const double shift, const double* __restrict__ b,
                                                             double* __restrict__ x, double* __restrict__ r, const PcgBufs w)
{
    __shared__ double red[4];
    __shared__ double rloc[kRowsPerWg * 6];
    const uint32_t n = A.n_rows * 6;
    const uint32_t row0 = blockIdx.x * kRowsPerWg * 6;
    const int t = threadIdx.x;
    const uint32_t i = row0 + t;
    const bool act = t < kRowsPerWg * 6 && i < n;
    if (t < kRowsPerWg) {
        const uint32_t f = blockIdx.x * kRowsPerWg + t;
        if (f < A.n_rows && !spd6_inverse_shift(A.blocks + 36 * (size_t)A.diag_block[f], shift, w.minv + 36 * (size_t)f))
            w.state[ST_BREAKDOWN] = 1.0;
    }
    double bi = 0.0;
    if (act) { bi = b[i]; x[i] = 0.0; r[i] = bi; rloc[t] = bi; }
    __syncthreads();
    double rz = 0.0;
    if (act) {
        const double* M = w.minv + 36 * (size_t)(i / 6) + 6 * (i % 6);
        const double* rv = rloc + (t / 6) * 6;
        const double zi = M[0] * rv[0] + M[1] * rv[1] + M[2] * rv[2] + M[3] * rv[3] + M[4] * rv[4] + M[5] * rv[5];
        w.z[i] = zi;
        w.p[0][i] = zi;
        rz = bi * zi;
    }
    const double bb = block_sum256(bi * bi, red);
    rz = block_sum256(rz, red);
    if (t == 0) {
        w.part_rr[blockIdx.x] = bb;
        w.part_rz[blockIdx.x] = rz;
        if (blockIdx.x == 0) { w.state[ST_ITERS] = 0.0; }
    }
}

// |b|^2 is the parity-0 rr of the init kernel; keep a copy that survives the iterations
__global__ void pcgm_bb_kernel(const PcgBufs w)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) w.state[ST_BB] = sum_partials(w.part_rr, w.n_wg);
}

// Sum of the per-workgroup partials by the whole workgroup: lane t adds every 256th value, then the fixed tree of
// block_sum256.  Every workgroup runs the same code on the same data, so all of them get the same bits - and none
// walks the list alone (120 dependent-latency loads per lane at 5000 block rows was most of an iteration).
__device__ __forceinline__ double coop_sum(const double* __restrict__ part, uint32_t n, double* red)
{
    double v = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += kThreads) v += part[i];
    return block_sum256(v, red);
}

// Three such sums at once: the loads of all three lists are in flight together and the workgroup synchronises twice
// instead of six times.  The per-lane subsequences and the tree are those of coop_sum: same bits.
__device__ __forceinline__ void coop_sum3(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                                          uint32_t n, double (*red3)[4], double& sa, double& sb, double& sc)
{
    double va = 0.0, vb = 0.0, vc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += kThreads) { va += a[i]; vb += b[i]; vc += c[i]; }
    va = wave_sum(va); vb = wave_sum(vb); vc = wave_sum(vc);
    __syncthreads();
    if (threadIdx.x % 64 == 0) { red3[0][threadIdx.x / 64] = va; red3[1][threadIdx.x / 64] = vb; red3[2][threadIdx.x / 64] = vc; }
    __syncthreads();
    sa = red3[0][0] + red3[0][1] + red3[0][2] + red3[0][3];
    sb = red3[1][0] + red3[1][1] + red3[1][2] + red3[1][3];
    sc = red3[2][0] + red3[2][1] + red3[2][2] + red3[2][3];
}

// q_row of (A p) with p = z + beta p_old formed on the fly (FIRST: p is p_old itself, the init kernel's z)
// (sub, lanes): this lane's share of the row - entries sub, sub + lanes, ... - for callers that put several lanes on a row and add
// their sums (pcg2_matvec on small aggregates)
template <bool FIRST>
__device__ __forceinline__ double bsr_row_dot_dir(const BsrView& A, uint32_t row, const double* __restrict__ z,
                                                  const double* __restrict__ p_old, double beta, const uint32_t sub = 0, const uint32_t lanes = 1)
{
    const uint32_t f = row / 6, a = row % 6;
    double s = 0.0;
    // Eight entries of the row at a time: their indices first, then their blocks and operand pieces - two rounds of loads for eight
    // entries instead of two per entry (a row of the pose graph's H has ~9 blocks, of a wide-band S ~39: the loop was a chain of
    // dependent gathers, 10.5 us per launch at configs[4]).  The products are added in the row's order: the bits do not change.
    constexpr int kRowUnroll = 8;
    const uint32_t e_end = A.row_ptr[f + 1];
    for (uint32_t e0 = A.row_ptr[f] + sub; e0 < e_end; e0 += kRowUnroll * lanes) {
        uint32_t blk[kRowUnroll], col[kRowUnroll];
        bool tr[kRowUnroll], on[kRowUnroll];
#pragma unroll
        for (int u = 0; u < kRowUnroll; u++) {
            on[u] = e0 + u * lanes < e_end;
            const uint32_t e = on[u] ? e0 + u * lanes : e0;
            blk[u] = A.ent_blk[e]; col[u] = A.ent_col[e]; tr[u] = A.ent_trans[e] != 0;
        }
        double bv[kRowUnroll][6], xv[kRowUnroll][6];
#pragma unroll
        for (int u = 0; u < kRowUnroll; u++) {
            const double* B = A.blocks + 36 * (size_t)blk[u];
            const size_t c0 = 6 * (size_t)col[u];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                bv[u][c] = tr[u] ? B[c * 6 + a] : B[a * 6 + c];
                xv[u][c] = FIRST ? p_old[c0 + c] : z[c0 + c] + beta * p_old[c0 + c];
            }
        }
#pragma unroll
        for (int u = 0; u < kRowUnroll; u++)
            if (on[u]) {
#pragma unroll
                for (int c = 0; c < 6; c++) s += bv[u][c] * xv[u][c];
            }
    }
    return s;
}

// parity = iteration index & 1: the sums of the previous update (or of the init kernel) sit at `parity`, the ones before
// at parity ^ 1; p of this iteration goes to w.p[parity], the previous one is w.p[parity ^ 1] (FIRST: w.p[0] is ready)
template <bool FIRST>
__global__ __launch_bounds__(kThreads) void pcgm_matvec_kernel(const BsrView A, const double shift, const PcgBufs w,
                                                               const int parity, const double tol)
{
    __shared__ double red[4];
    __shared__ double red3[3][4];
    if (w.state[ST_DONE] != 0.0) return;   // uniform: written by an earlier launch
    double rr, rz_new, rz_old;
    coop_sum3(w.part_rr + parity * w.n_wg, w.part_rz + parity * w.n_wg, w.part_rz + (parity ^ 1) * w.n_wg, w.n_wg, red3, rr, rz_new, rz_old);
    // the previous update may have reached the tolerance: remember it (ST_DONE), or every second launch of the rest of
    // the chunk would iterate on the stale sums of the other parity
    if (!(rr > tol * tol * w.state[ST_BB]) || w.state[ST_BREAKDOWN] != 0.0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) w.state[ST_DONE] = 1.0;
        return;
    }
    const double beta = FIRST ? 0.0 : rz_new / rz_old;
    const double* p_old = w.p[FIRST ? 0 : parity ^ 1];
    const uint32_t n = A.n_rows * 6;
    const int t = threadIdx.x;
    const uint32_t i = blockIdx.x * kRowsPerWg * 6 + t;
    double pq = 0.0;
    if (t < kRowsPerWg * 6 && i < n) {
        const double pi = FIRST ? p_old[i] : w.z[i] + beta * p_old[i];
        if (!FIRST) w.p[parity][i] = pi;
        const double qi = bsr_row_dot_dir<FIRST>(A, i, w.z, p_old, beta) + shift * pi;
        w.q[i] = qi;
        pq = pi * qi;
    }
    pq = block_sum256(pq, red);
    if (t == 0) w.part_pq[blockIdx.x] = pq;
}

__global__ __launch_bounds__(kThreads) void pcgm_update_kernel(const uint32_t n, double* __restrict__ x, double* __restrict__ r,
                                                               const PcgBufs w, const int parity, const double tol)
{
    __shared__ double red[4];
    __shared__ double red3[3][4];
    __shared__ double rloc[kRowsPerWg * 6];
    if (w.state[ST_DONE] != 0.0) return;   // uniform: written by an earlier launch
    double rr_old, pq, rz;
    coop_sum3(w.part_rr + parity * w.n_wg, w.part_pq, w.part_rz + parity * w.n_wg, w.n_wg, red3, rr_old, pq, rz);
    if (!(rr_old > tol * tol * w.state[ST_BB]) || w.state[ST_BREAKDOWN] != 0.0) return;   // as the matvec of this iteration decided
    const int t = threadIdx.x;
    if (!(pq > 0.0)) {
        if (t == 0 && blockIdx.x == 0) w.state[ST_BREAKDOWN] = 2.0;
        return;
    }
    const double alpha = rz / pq;
    const uint32_t i = blockIdx.x * kRowsPerWg * 6 + t;
    const bool act = t < kRowsPerWg * 6 && i < n;
    double ri = 0.0;
    if (act) {
        x[i] += alpha * w.p[parity][i];
        ri = r[i] - alpha * w.q[i];
        r[i] = ri;
        rloc[t] = ri;
    }
    __syncthreads();
    double rzn = 0.0;
    if (act) {
        const double* M = w.minv + 36 * (size_t)(i / 6) + 6 * (i % 6);
        const double* rv = rloc + (t / 6) * 6;
        const double zi = M[0] * rv[0] + M[1] * rv[1] + M[2] * rv[2] + M[3] * rv[3] + M[4] * rv[4] + M[5] * rv[5];
        w.z[i] = zi;
        rzn = ri * zi;
    }
    const double rr = block_sum256(ri * ri, red);
    rzn = block_sum256(rzn, red);
    if (t == 0) {
        w.part_rr[(parity ^ 1) * w.n_wg + blockIdx.x] = rr;
        w.part_rz[(parity ^ 1) * w.n_wg + blockIdx.x] = rzn;
        if (blockIdx.x == 0) w.state[ST_ITERS] += 1.0;
    }
}


// ---- two-level PCG: block-Jacobi + an aggregation coarse space ------------------------------------------------------------------
// For the pose graph (pg_solver.hip): the graph's Hessian has smooth near-null modes - rigid motions of whole regions - that
// block-Jacobi sees one vertex at a time (850 iterations per solve on BASELINE.json configs[4]).  The additive two-level
// preconditioner   z = D^-1 r + P (P^T (A + shift I) P)^-1 P^T r   adds a coarse space of six rigid-body modes per aggregate
// of at most 42 neighbouring vertices (P_i = the vertex's local increment under a rigid motion of its aggregate; the
// coarse matrix is inverted explicitly by launch_dense_spd_inverse60, so applying it is one small dense product).  A CPU
// prototype on configs[4] gave 368 iterations for block-Jacobi, 46 - 75 for this (aggregates of 25 - 100 vertices).
// One workgroup per aggregate (the block rows of an aggregate are contiguous): TWO kernels per iteration -
//   matvec   as pcgm_matvec, plus this aggregate's P^T q;
//   update   alpha, x, r, the fine part D^-1 r; P^T r of ALL aggregates by its recurrence P^T r - alpha P^T q (every workgroup
//            for itself - what made the coarse part a third kernel was waiting for the other aggregates' P^T r);
//            y = (row block of the coarse inverse) . (P^T r), z = D^-1 r + P y, partial r.r and r.z.
struct Pcg2Bufs {
    double *p[2], *z, *q, *minv, *part_pq, *part_rz, *part_rr, *state;
    double *rc[2], *qc;   // P^T r of every aggregate by iteration parity, P^T q of the current iteration (ncp entries each)
    uint32_t n_wg;
};

__global__ __launch_bounds__(kThreads) void pcg2_init_kernel(const BsrView A, const double shift, const double* __restrict__ b,
                                                             double* __restrict__ x, double* __restrict__ r, const Pcg2Bufs w,
                                                             const TwoLevelView tl)
{
    __shared__ double red[4];
    __shared__ double rloc[kRowsPerWg * 6], yl[kRowsPerWg * 6];
    const uint32_t r0 = tl.agg_ptr[blockIdx.x], nr = tl.agg_ptr[blockIdx.x + 1] - r0;
    const int t = threadIdx.x;
    const uint32_t i = r0 * 6 + t;
    const bool act = t < (int)(nr * 6);
    if (t < (int)nr) {
        const uint32_t f = r0 + t;
        if (!spd6_inverse_shift(A.blocks + 36 * (size_t)A.diag_block[f], shift, w.minv + 36 * (size_t)f)) w.state[ST_BREAKDOWN] = 1.0;
    }
    double bi = 0.0;
    if (act) { bi = b[i]; x[i] = 0.0; r[i] = bi; rloc[t] = bi; }
    __syncthreads();
    if (act) {
        const double* M = w.minv + 36 * (size_t)(i / 6) + 6 * (i % 6);
        const double* rv = rloc + (t / 6) * 6;
        w.z[i] = M[0] * rv[0] + M[1] * rv[1] + M[2] * rv[2] + M[3] * rv[3] + M[4] * rv[4] + M[5] * rv[5];
        // (P_v^T r_v)[a], a = t % 6
        const double* Pv = tl.P + 36 * (size_t)(i / 6);
        const int a = t % 6;
        yl[t] = Pv[a] * rv[0] + Pv[6 + a] * rv[1] + Pv[12 + a] * rv[2] + Pv[18 + a] * rv[3] + Pv[24 + a] * rv[4] + Pv[30 + a] * rv[5];
    }
    __syncthreads();
    if (t < 6) {
        double s = 0.0;
        for (uint32_t v = 0; v < nr; v++) s += yl[v * 6 + t];
        w.rc[0][blockIdx.x * 6 + t] = s;
    }
    const double bb = block_sum256(bi * bi, red);
    if (t == 0) {
        w.part_rr[blockIdx.x] = bb;
        if (blockIdx.x == 0) w.state[ST_ITERS] = 0.0;
    }
}

// INIT: behind pcg2_init (writes p[0] = z and the parity-0 r.z); else behind the update of iteration `parity`
template <bool INIT>
__global__ __launch_bounds__(kThreads) void pcg2_coarse_kernel(const double* __restrict__ r, const Pcg2Bufs w, const TwoLevelView tl,
                                                               const int parity, const double tol)
{
    __shared__ double red[4];
    __shared__ double rcl[1280];          // P^T r of every aggregate (ncp <= 1260)
    __shared__ double part[6][40];
    __shared__ double y[6];
    if (!INIT) {
        if (w.state[ST_DONE] != 0.0 || w.state[ST_BREAKDOWN] != 0.0) return;
        const double rr_old = coop_sum(w.part_rr + parity * w.n_wg, w.n_wg, red);
        if (!(rr_old > tol * tol * w.state[ST_BB])) return;   // as the update of this iteration decided
        __syncthreads();
    }
    const int t = threadIdx.x;
    const uint32_t r0 = tl.agg_ptr[blockIdx.x], nr = tl.agg_ptr[blockIdx.x + 1] - r0;
    for (uint32_t j = t; j < tl.ncp; j += kThreads) rcl[j] = w.rc[0][j];
    __syncthreads();
    if (t < 240) {
        const int a = t / 40, s = t % 40;
        const double* row = tl.Ainv + (size_t)(blockIdx.x * 6 + a) * tl.ncp;
        double sum = 0.0;
        for (uint32_t j = s; j < tl.ncp; j += 40) sum += row[j] * rcl[j];
        part[a][s] = sum;
    }
    __syncthreads();
    if (t < 6) {
        double sum = 0.0;
#pragma unroll
        for (int s = 0; s < 40; s++) sum += part[t][s];
        y[t] = sum;
    }
    __syncthreads();
    const uint32_t i = r0 * 6 + t;
    double rz = 0.0;
    if (t < (int)(nr * 6)) {
        const double* Pv = tl.P + 36 * (size_t)(i / 6) + 6 * (i % 6);
        const double zi = w.z[i] + (Pv[0] * y[0] + Pv[1] * y[1] + Pv[2] * y[2] + Pv[3] * y[3] + Pv[4] * y[4] + Pv[5] * y[5]);
        w.z[i] = zi;
        if (INIT) w.p[0][i] = zi;
        rz = r[i] * zi;
    }
    rz = block_sum256(rz, red);
    if (t == 0) w.part_rz[(INIT ? 0 : (parity ^ 1)) * w.n_wg + blockIdx.x] = rz;
}

template <bool FIRST>
__global__ __launch_bounds__(kThreads) void pcg2_matvec_kernel(const BsrView A, const double shift, const Pcg2Bufs w, const TwoLevelView tl,
                                                               const int parity, const double tol)
{
    __shared__ double red[4];
    __shared__ double red3[3][4];
    if (w.state[ST_DONE] != 0.0) return;
    double rr, rz_new, rz_old;
    coop_sum3(w.part_rr + parity * w.n_wg, w.part_rz + parity * w.n_wg, w.part_rz + (parity ^ 1) * w.n_wg, w.n_wg, red3, rr, rz_new, rz_old);
    if (!(rr > tol * tol * w.state[ST_BB]) || w.state[ST_BREAKDOWN] != 0.0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) w.state[ST_DONE] = 1.0;
        return;
    }
    const double beta = FIRST ? 0.0 : rz_new / rz_old;
    const double* p_old = w.p[FIRST ? 0 : parity ^ 1];
    const uint32_t r0 = tl.agg_ptr[blockIdx.x], nr = tl.agg_ptr[blockIdx.x + 1] - r0;
    const int t = threadIdx.x;
    const uint32_t i = r0 * 6 + t;
    __shared__ double qloc[kRowsPerWg * 6], yl[kRowsPerWg * 6];
    double pq = 0.0;
    const bool act = t < (int)(nr * 6);
    // A small aggregate (the BA path's six cameras: 36 rows of ~39 blocks each) leaves most of the workgroup idle with a lane per row:
    // four (or two) lanes share a row's blocks and add their sums by DPP
    const uint32_t rows = nr * 6;
    const uint32_t L = rows * 4 <= (uint32_t)kThreads ? 4u : (rows * 2 <= (uint32_t)kThreads ? 2u : 1u);   // uniform in the workgroup
    {
        const uint32_t row_l = (uint32_t)t / L, sub = (uint32_t)t % L;
        const bool on = row_l < rows;
        const uint32_t ir = r0 * 6 + (on ? row_l : 0u);
        double part = on ? bsr_row_dot_dir<FIRST>(A, ir, w.z, p_old, beta, sub, L) : 0.0;
        if (L == 4) part = quad_sum(part);
        else if (L == 2) part += mov_dpp_f64<kDppXor1>(part);
        if (on && sub == 0) {
            const double pi = FIRST ? p_old[ir] : w.z[ir] + beta * p_old[ir];
            if (!FIRST) w.p[parity][ir] = pi;
            const double qi = part + shift * pi;
            w.q[ir] = qi;
            qloc[row_l] = qi;
            pq = pi * qi;
        }
    }
    pq = block_sum256(pq, red);   // (its barriers: qloc is complete behind it)
    if (t == 0) w.part_pq[blockIdx.x] = pq;
    // P^T q of this aggregate: the update kernels of ALL aggregates advance P^T r with it
    if (act) {
        const double* Pv = tl.P + 36 * (size_t)(i / 6);
        const double* qv = qloc + (t / 6) * 6;
        const int a = t % 6;
        yl[t] = Pv[a] * qv[0] + Pv[6 + a] * qv[1] + Pv[12 + a] * qv[2] + Pv[18 + a] * qv[3] + Pv[24 + a] * qv[4] + Pv[30 + a] * qv[5];
    }
    __syncthreads();
    if (t < 6) {
        double sq = 0.0;
        for (uint32_t v = 0; v < nr; v++) sq += yl[v * 6 + t];
        w.qc[blockIdx.x * 6 + t] = sq;
    }
}

__global__ __launch_bounds__(kThreads) void pcg2_update_kernel(double* __restrict__ x, double* __restrict__ r, const Pcg2Bufs w,
                                                               const TwoLevelView tl, const int parity, const double tol)
{
    __shared__ double red[4];
    __shared__ double red3[3][4];
    __shared__ double rloc[kRowsPerWg * 6];
    __shared__ double rcl[1280];          // P^T r of every aggregate (ncp <= 1260)
    __shared__ double part[6][40];
    __shared__ double y[6];
    if (w.state[ST_DONE] != 0.0) return;
    double rr_old, pq, rz;
    coop_sum3(w.part_rr + parity * w.n_wg, w.part_pq, w.part_rz + parity * w.n_wg, w.n_wg, red3, rr_old, pq, rz);
    if (!(rr_old > tol * tol * w.state[ST_BB]) || w.state[ST_BREAKDOWN] != 0.0) return;
    const int t = threadIdx.x;
    if (!(pq > 0.0)) {
        if (t == 0 && blockIdx.x == 0) w.state[ST_BREAKDOWN] = 2.0;
        return;
    }
    const double alpha = rz / pq;
    const uint32_t r0 = tl.agg_ptr[blockIdx.x], nr = tl.agg_ptr[blockIdx.x + 1] - r0;
    const uint32_t i = r0 * 6 + t;
    const bool act = t < (int)(nr * 6);
    double ri = 0.0;
    if (act) {
        x[i] += alpha * w.p[parity][i];
        ri = r[i] - alpha * w.q[i];
        r[i] = ri;
        rloc[t] = ri;
    }
    // P^T r of every aggregate: the recurrence (P^T is linear in r), this workgroup's own six entries handed on
    for (uint32_t j = t; j < tl.ncp; j += kThreads) {
        const double v = w.rc[parity][j] - alpha * w.qc[j];
        rcl[j] = v;
        if (j / 6 == blockIdx.x) w.rc[parity ^ 1][j] = v;
    }
    __syncthreads();
    double zf = 0.0;
    if (act) {
        const double* M = w.minv + 36 * (size_t)(i / 6) + 6 * (i % 6);
        const double* rv = rloc + (t / 6) * 6;
        zf = M[0] * rv[0] + M[1] * rv[1] + M[2] * rv[2] + M[3] * rv[3] + M[4] * rv[4] + M[5] * rv[5];   // the fine part D^-1 r
    }
    if (t < 240) {
        const int a = t / 40, sgrp = t % 40;
        const double* row = tl.Ainv + (size_t)(blockIdx.x * 6 + a) * tl.ncp;
        double sum = 0.0;
        for (uint32_t j = sgrp; j < tl.ncp; j += 40) sum += row[j] * rcl[j];
        part[a][sgrp] = sum;
    }
    __syncthreads();
    if (t < 6) {
        double sum = 0.0;
#pragma unroll
        for (int sgrp = 0; sgrp < 40; sgrp++) sum += part[t][sgrp];
        y[t] = sum;
    }
    __syncthreads();
    double rzn = 0.0;
    if (act) {
        const double* Pv = tl.P + 36 * (size_t)(i / 6) + 6 * (i % 6);
        const double zi = zf + (Pv[0] * y[0] + Pv[1] * y[1] + Pv[2] * y[2] + Pv[3] * y[3] + Pv[4] * y[4] + Pv[5] * y[5]);
        w.z[i] = zi;
        rzn = ri * zi;
    }
    const double rr = block_sum256(ri * ri, red);
    rzn = block_sum256(rzn, red);
    if (t == 0) {
        w.part_rr[(parity ^ 1) * w.n_wg + blockIdx.x] = rr;
        w.part_rz[(parity ^ 1) * w.n_wg + blockIdx.x] = rzn;
        if (blockIdx.x == 0) w.state[ST_ITERS] += 1.0;
    }
}

// G_I = sum over the aggregate's block rows of P_v^T P_v (the shift's share of the coarse matrix); one wave per aggregate
__global__ __launch_bounds__(64) void coarse_gram_kernel(const uint32_t* __restrict__ agg_ptr, const double* __restrict__ P, double* __restrict__ G)
{
    const uint32_t I = blockIdx.x;
    const int e = threadIdx.x;
    if (e >= 36) return;
    const int a = e / 6, b = e % 6;
    double s = 0.0;
    for (uint32_t f = agg_ptr[I]; f < agg_ptr[I + 1]; f++) {
        const double* Pv = P + 36 * (size_t)f;
#pragma unroll
        for (int c = 0; c < 6; c++) s += Pv[c * 6 + a] * Pv[c * 6 + b];
    }
    G[36 * (size_t)I + e] = s;
}

// one coarse block (I, J) of P^T H P: the sum over its fine blocks (entry = 2 * block + swap; swap: the stored block is H(i, j)
// with i in J and j in I, so its transpose is what couples I to J).  One workgroup per coarse block: seven groups of 36 lanes take
// every seventh fine block of the list, their partial sums are added in group order (fixed order: reproducible).
constexpr int kAsmGroups = 7;
__global__ __launch_bounds__(256) void coarse_assemble_kernel(const uint32_t* __restrict__ cb_ptr, const uint32_t* __restrict__ cb_ent,
                                                                 const uint32_t* __restrict__ cb_I, const uint32_t* __restrict__ cb_J,
                                                                 const uint32_t* __restrict__ blk_row, const uint32_t* __restrict__ blk_col,
                                                                 const double* __restrict__ H, const double* __restrict__ P, const uint32_t ncp,
                                                                 double* __restrict__ Ac0)
{
    __shared__ double T[kAsmGroups][36];
    __shared__ double part[kAsmGroups][36];
    const uint32_t cb = blockIdx.x;
    const int grp = threadIdx.x / 36, e = threadIdx.x % 36, a = e / 6, b = e % 6;
    const bool on = grp < kAsmGroups;
    double acc = 0.0;
    const uint32_t q0 = cb_ptr[cb], q1 = cb_ptr[cb + 1];
    // uniform trip count: every lane reaches the barriers
    for (uint32_t base = q0; base < q1; base += kAsmGroups) {
        const uint32_t q = base + (uint32_t)grp;
        const bool live = on && q < q1;
        const double* L = nullptr;
        if (live) {
            const uint32_t blk = cb_ent[q] >> 1, swap = cb_ent[q] & 1;
            const uint32_t i = blk_row[blk], j = blk_col[blk];
            const double* Hb = H + 36 * (size_t)blk;
            L = P + 36 * (size_t)(swap ? j : i);
            const double* Rr = P + 36 * (size_t)(swap ? i : j);
            double t = 0.0;   // T = Hm Rr with Hm = Hb or Hb^T
#pragma unroll
            for (int d = 0; d < 6; d++) t += (swap ? Hb[d * 6 + a] : Hb[a * 6 + d]) * Rr[d * 6 + b];
            T[grp][e] = t;
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int c = 0; c < 6; c++) acc += L[c * 6 + a] * T[grp][c * 6 + b];
        }
        __syncthreads();
    }
    if (on) part[grp][e] = acc;
    __syncthreads();
    if (threadIdx.x < 36) {
        double sum = 0.0;
#pragma unroll
        for (int g = 0; g < kAsmGroups; g++) sum += part[g][threadIdx.x];
        Ac0[(size_t)(6 * cb_I[cb] + a) * ncp + 6 * cb_J[cb] + b] = sum;
    }
}


// Ainv <- Ac0 + shift * blockdiag(G) on the coarse unknowns, the identity on the padding; the caller inverts it in place
__global__ __launch_bounds__(256) void pcg2_coarse_matrix_kernel(const double* __restrict__ Ac0, const double* __restrict__ G, const double shift,
                                                                 const uint32_t nc, const uint32_t ncp, double* __restrict__ out)
{
    const uint32_t row = blockIdx.x;
    for (uint32_t col = threadIdx.x; col < ncp; col += 256) {
        double v;
        if (row >= nc || col >= nc) v = row == col ? 1.0 : 0.0;
        else {
            v = Ac0[(size_t)row * ncp + col];
            if (row / 6 == col / 6) v += shift * G[36 * (size_t)(row / 6) + (row % 6) * 6 + col % 6];
        }
        out[(size_t)row * ncp + col] = v;
    }
}

}  // namespace

size_t pcg_multi_work_count(uint32_t n_rows)
{
    const size_t n = (size_t)n_rows * 6, n_wg = (n_rows + kRowsPerWg - 1) / kRowsPerWg;
    return 4 * n + 36 * (size_t)n_rows + 5 * n_wg + ST_COUNT + 16;
}

int pcg_multi_solve(hipStream_t s, const BsrView& A, double shift, const double* b, double* x, double* resid,
                    double* work, double tol, int max_iter, int chunk, double* rel_resid)
{
    if (rel_resid) *rel_resid = 0.0;
    if (!A.n_rows) return 0;
    const uint32_t n = A.n_rows * 6, n_wg = (A.n_rows + kRowsPerWg - 1) / kRowsPerWg;
    PcgBufs w;
    w.n_wg = n_wg;
    w.p[0] = work; w.p[1] = w.p[0] + n; w.z = w.p[1] + n; w.q = w.z + n; w.minv = w.q + n;
    w.part_pq = w.minv + 36 * (size_t)A.n_rows;
    w.part_rz = w.part_pq + n_wg;
    w.part_rr = w.part_rz + 2 * (size_t)n_wg;
    w.state = w.part_rr + 2 * (size_t)n_wg;
    if (!resid) return -2;   // the residual vector is part of the iteration state
    double* r = resid;
    (void)hipMemsetAsync(w.state, 0, sizeof(double) * ST_COUNT, s);
    hipLaunchKernelGGL(pcgm_init_kernel, dim3(n_wg), dim3(kThreads), 0, s, A, shift, b, x, r, w);
    hipLaunchKernelGGL(pcgm_bb_kernel, dim3(1), dim3(64), 0, s, w);
    int parity = 0, launched = 0;
    double host_state[ST_COUNT] = {0, 0, 0, 0};
    double rr = 0.0;
    while (launched < max_iter) {
        const int todo = std::min(chunk, max_iter - launched);
        for (int k = 0; k < todo; k++) {
            if (launched + k == 0) hipLaunchKernelGGL(pcgm_matvec_kernel<true>, dim3(n_wg), dim3(kThreads), 0, s, A, shift, w, parity, tol);
            else hipLaunchKernelGGL(pcgm_matvec_kernel<false>, dim3(n_wg), dim3(kThreads), 0, s, A, shift, w, parity, tol);
            hipLaunchKernelGGL(pcgm_update_kernel, dim3(n_wg), dim3(kThreads), 0, s, n, x, r, w, parity, tol);
            parity ^= 1;
        }
        launched += todo;
        // one round trip per chunk: iterations done, |b|^2, breakdown flag, and the current |r|^2 partials
        std::vector<double> rrp(2 * (size_t)n_wg);
        if (hipMemcpyAsync(host_state, w.state, sizeof host_state, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipMemcpyAsync(rrp.data(), w.part_rr, sizeof(double) * rrp.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        if (host_state[ST_BREAKDOWN] != 0.0) return -1;
        const int done = (int)host_state[ST_ITERS];
        const int par = done & 1;   // parity after `done` updates
        rr = 0.0;
        for (uint32_t i = 0; i < n_wg; i++) rr += rrp[(size_t)par * n_wg + i];
        if (!(rr > tol * tol * host_state[ST_BB]) || done < launched) {
            if (rel_resid) *rel_resid = host_state[ST_BB] > 0 ? std::sqrt(rr / host_state[ST_BB]) : 0.0;
            return done;
        }
    }
    if (rel_resid) *rel_resid = host_state[ST_BB] > 0 ? std::sqrt(rr / host_state[ST_BB]) : 0.0;
    return (int)host_state[ST_ITERS];
}

}  // namespace soslam

namespace soslam {

size_t pcg2_work_count(uint32_t n_rows, uint32_t n_agg)
{
    const size_t n = (size_t)n_rows * 6;
    return 4 * n + 36 * (size_t)n_rows + 5 * (size_t)n_agg + ST_COUNT + 16 + 3 * 1280;
}

// the coarse operator for a shift: (P^T A P + shift P^T P)^-1, explicit, into ainv_out ([ncp * ncp]); status[0] = 1 on a
// non-positive pivot tile.  Its own call so that the caller may run it on another stream for a LATER solve (pg_solver.hip).
void pcg2_coarse_inverse(hipStream_t s, uint32_t n_agg, uint32_t ncp, const double* Ac0, const double* G, double shift, double* ainv_out,
                         double* ebuf, double* status)
{
    if (!n_agg) return;
    hipLaunchKernelGGL(pcg2_coarse_matrix_kernel, dim3(ncp), dim3(256), 0, s, Ac0, G, shift, n_agg * 6, ncp, ainv_out);
    launch_dense_spd_inverse60(s, ainv_out, (int)(ncp / 60), ebuf, status);
}

int pcg2_solve(hipStream_t s, const BsrView& A, double shift, const double* b, double* x, double* resid, double* work, const TwoLevelView& tl,
               double tol, int max_iter, int chunk, double* rel_resid)
{
    if (rel_resid) *rel_resid = 0.0;
    if (!A.n_rows) return 0;
    if (!resid || tl.ncp > 1260 || tl.ncp % 60 != 0) return -2;
    const uint32_t n = A.n_rows * 6, n_wg = tl.n_agg;
    Pcg2Bufs w;
    w.n_wg = n_wg;
    w.p[0] = work; w.p[1] = w.p[0] + n; w.z = w.p[1] + n; w.q = w.z + n; w.minv = w.q + n;
    w.part_pq = w.minv + 36 * (size_t)A.n_rows;
    w.part_rz = w.part_pq + n_wg;
    w.part_rr = w.part_rz + 2 * (size_t)n_wg;
    w.state = w.part_rr + 2 * (size_t)n_wg;
    w.rc[0] = w.state + ST_COUNT; w.rc[1] = w.rc[0] + 1280; w.qc = w.rc[1] + 1280;
    double* r = resid;
    (void)hipMemsetAsync(w.state, 0, sizeof(double) * (ST_COUNT + 3 * 1280), s);   // (the padding entries of P^T r and P^T q stay zero)
    hipLaunchKernelGGL(pcg2_init_kernel, dim3(n_wg), dim3(kThreads), 0, s, A, shift, b, x, r, w, tl);
    hipLaunchKernelGGL(pcgm_bb_kernel, dim3(1), dim3(64), 0, s, PcgBufs{{w.p[0], w.p[1]}, w.z, w.q, w.minv, w.part_pq, w.part_rz, w.part_rr, w.state, w.n_wg});
    hipLaunchKernelGGL(pcg2_coarse_kernel<true>, dim3(n_wg), dim3(kThreads), 0, s, r, w, tl, 0, tol);
    int parity = 0, launched = 0;
    double host_state[ST_COUNT] = {0, 0, 0, 0};
    double rr = 0.0;
    while (launched < max_iter) {
        // the first chunk is the caller's estimate of the whole solve (the iterations its last solve took); what is left
        // after it goes in short chunks: an unneeded launch costs 2 - 3 us, two per iteration
        const int todo = std::min(launched == 0 ? std::max(chunk, 1) : 8, max_iter - launched);
        for (int k = 0; k < todo; k++) {
            if (launched + k == 0) hipLaunchKernelGGL(pcg2_matvec_kernel<true>, dim3(n_wg), dim3(kThreads), 0, s, A, shift, w, tl, parity, tol);
            else hipLaunchKernelGGL(pcg2_matvec_kernel<false>, dim3(n_wg), dim3(kThreads), 0, s, A, shift, w, tl, parity, tol);
            hipLaunchKernelGGL(pcg2_update_kernel, dim3(n_wg), dim3(kThreads), 0, s, x, r, w, tl, parity, tol);
            parity ^= 1;
        }
        launched += todo;
        std::vector<double> rrp(2 * (size_t)n_wg);
        if (hipMemcpyAsync(host_state, w.state, sizeof host_state, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipMemcpyAsync(rrp.data(), w.part_rr, sizeof(double) * rrp.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        if (host_state[ST_BREAKDOWN] != 0.0) return -1;
        const int done = (int)host_state[ST_ITERS];
        const int par = done & 1;
        rr = 0.0;
        for (uint32_t i = 0; i < n_wg; i++) rr += rrp[(size_t)par * n_wg + i];
        if (!(rr > tol * tol * host_state[ST_BB]) || done < launched) {
            if (rel_resid) *rel_resid = host_state[ST_BB] > 0 ? std::sqrt(rr / host_state[ST_BB]) : 0.0;
            return done;
        }
    }
    if (rel_resid) *rel_resid = host_state[ST_BB] > 0 ? std::sqrt(rr / host_state[ST_BB]) : 0.0;
    return (int)host_state[ST_ITERS];
}

// G = P^T P per aggregate and Ac0 = P^T A P (dense, leading dimension ncp) from the block rows' coarse bases P and the blocks of A
void launch_coarse_assemble(hipStream_t s, uint32_t n_agg, const uint32_t* agg_ptr, uint32_t n_cb, const uint32_t* cb_ptr, const uint32_t* cb_ent,
                            const uint32_t* cb_I, const uint32_t* cb_J, const uint32_t* blk_row, const uint32_t* blk_col, const double* blocks,
                            const double* P, uint32_t ncp, double* G, double* Ac0)
{
    if (!n_agg) return;
    hipLaunchKernelGGL(coarse_gram_kernel, dim3(n_agg), dim3(64), 0, s, agg_ptr, P, G);
    (void)hipMemsetAsync(Ac0, 0, sizeof(double) * (size_t)ncp * ncp, s);
    if (n_cb) hipLaunchKernelGGL(coarse_assemble_kernel, dim3(n_cb), dim3(256), 0, s, cb_ptr, cb_ent, cb_I, cb_J, blk_row, blk_col, blocks, P, ncp, Ac0);
}

// host: the non-empty coarse blocks (ordered aggregate pairs) of a block-sparse matrix (upper blocks blk_row <= blk_col) and the
// fine blocks that feed them, in block order (entry = 2 * block + swap)
void two_level_lists(uint32_t n_agg, const uint32_t* row_agg, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col,
                     std::vector<uint32_t>& cb_ptr, std::vector<uint32_t>& cb_ent, std::vector<uint32_t>& cb_I, std::vector<uint32_t>& cb_J)
{
    std::vector<std::pair<uint64_t, uint32_t>> items;
    items.reserve((size_t)n_blocks * 2);
    for (uint32_t blk = 0; blk < n_blocks; blk++) {
        const uint64_t I = row_agg[blk_row[blk]], J = row_agg[blk_col[blk]];
        items.push_back({I * n_agg + J, 2 * blk});
        if (blk_row[blk] != blk_col[blk]) items.push_back({J * n_agg + I, 2 * blk + 1});
    }
    std::sort(items.begin(), items.end());
    cb_ptr.clear(); cb_ent.clear(); cb_I.clear(); cb_J.clear();
    for (size_t q = 0; q < items.size(); q++) {
        if (q == 0 || items[q].first != items[q - 1].first) {
            cb_ptr.push_back((uint32_t)q);
            cb_I.push_back((uint32_t)(items[q].first / n_agg));
            cb_J.push_back((uint32_t)(items[q].first % n_agg));
        }
        cb_ent.push_back(items[q].second);
    }
    cb_ptr.push_back((uint32_t)items.size());
}

}  // namespace soslam
