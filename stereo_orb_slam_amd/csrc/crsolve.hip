// crsolve.hip - block cyclic reduction for the block-banded reduced camera matrix (gfx950).
//
// A band of half-width bw blocks is block TRIDIAGONAL in super-blocks of bw cameras (sb = 6 bw scalars).  The
// sequential band Cholesky needs n dependent block steps on one CU (latency-bound: ~2 us per step, 1.2 ms at 499
// cameras).  Cyclic reduction instead eliminates every other super-block of the current level IN PARALLEL and halves
// the system per level: ceil(log2 m) levels of two kernels each (m = 56 super-blocks at BASELINE.json configs[2]).
//
// Level h, eliminated nodes i = h (2j+1), remaining nodes k = 2 h j, F_n = A(n, n-h) the coupling to the left:
//   cr_invert : E_i = D_i^-1                       (block Gauss-Jordan sweep, 3x3 pivots, matrix in the accumulator registers
//                                                   of the f64 matrix cores, one workgroup per node)
//   cr_reduce : per remaining k with a = k-h, c = k+h (f64 matrix cores, one workgroup per 16 columns of the node)
//                 Q_a = E_a F_k^T        P_c = E_c F_c
//                 D_k -= F_k Q_a + F_c^T P_c        F_k <- -Q_a^T F_a      (fill: A(k, k-2h), written to the other coupling array)
// Each product is owned by exactly one remaining node.  With the explicit inverse every step of a later solve is a
// matrix-vector product:   forward   w_k -= Q_a^T w_a + P_c^T w_c      backward   x_i = E_i w_i - P_i x_{i-h} - Q_i x_{i+h}
// i.e. one small launch per level and sweep - two levels per launch with cr_fwd2 / cr_bwd2 / cr_top2 (ceil(log2 m) - 1
// launches per solve: 5 at m = 56).  All sums run in a fixed
// order: bitwise reproducible.  Positive definiteness is inherited by every Schur complement, so the sweep needs no
// pivoting; a non-positive 3x3 pivot block raises scal[SC_LIN_STATUS].
#include "linsolve.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ba_kernels.h"
#include "reduce.h"

namespace soslam {

namespace {

constexpr int kCrReduceThreads = 256;

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int kLd = 65;   // leading dimension of the zero-padded LDS images fed to the matrix cores (sb <= 60): odd, so that
                          // transposed staging writes are conflict-free; 16 consecutive doubles of a row still cover every bank once

// One 16x16 tile of C = P^T Q on the f64 matrix cores: C[i0+i][j0+j] = sum_m P[m][i0+i] Q[m][j0+j], m < kp (multiple
// of 4), P and Q zero-padded LDS images with leading dimension kLd.  v_mfma_f64_16x16x4_f64 register layout
// (scripts/mfma_f64_layout.hip): lane l feeds A[i = l%16][k = l/16] and B[k = l/16][j = l%16] and receives
// D[i = 4 r + l/16][j = l%16] in accumulator register r.  Both operand reads are 16 consecutive doubles per k.
// The images have kImgRows = 64 rows (zero beyond sb), so the k loop has a fixed trip count and is fully unrolled: all
// operand reads of a tile are in flight before the first matrix instruction, which then issue back to back (a rolled loop
// paid an LDS round trip plus the accumulator's latency per step: 2 800 cycles per tile instead of 1 100).
constexpr int kImgRows = 64;
__device__ __forceinline__ double4_t mfma_ptq_tile(const double* __restrict__ P, const double* __restrict__ Q, int i0, int j0, int lane)
{
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = P + (lane / 16) * kLd + i0 + lane % 16;
    const double* pb = Q + (lane / 16) * kLd + j0 + lane % 16;
    double av[kImgRows / 4], bv[kImgRows / 4];
#pragma unroll
    for (int s = 0; s < kImgRows / 4; s++) { av[s] = pa[4 * s * kLd]; bv[s] = pb[4 * s * kLd]; }
#pragma unroll
    for (int s = 0; s < kImgRows / 4; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
    return acc;
}

struct CrView {
    uint32_t m;       // super-blocks
    uint32_t n_real;  // scalar rows of the system (the rest of the last super-block is padding)
    int bw, sb;       // cameras per super-block, scalars per super-block
    double *D, *F, *E, *P, *Q, *PT, *QT, *F2;   // each [m][sb*sb], row-major; P, Q (and transposes) indexed by the eliminated node;
                                                // F2: the couplings of the next level (cr_reduce reads one of F / F2 and writes the other)
};

// gather the upper block-sparse S into D (diagonal super-blocks) and F (F_i = A(i, i-1)) through a host-built map
// (cr_build_map): every element of both arrays is written - a 6x6 block of S, its transpose, zero, or the unit
// diagonal of a padding camera - so no memset precedes it.  blockIdx = (super-block, 0: D / 1: F).
__global__ __launch_bounds__(256) void cr_gather_kernel(double* __restrict__ blocks, const int32_t* __restrict__ map, const CrView v,
                                                        const CamDamp damp, const double* __restrict__ comp, const double comp_scale)
{
    const int sb = v.sb, bw = v.bw;
    const uint32_t i = blockIdx.x;
    const bool is_f = blockIdx.y != 0;
    if (damp.diagB && !is_f) {
        // the LM damping of this super-block's cameras (what ba_cam_damp_kernel does for the other solvers): every
        // diagonal block of S belongs to exactly one D workgroup, and no other workgroup reads it
        const uint32_t u = i * (uint32_t)sb + threadIdx.x;
        if ((int)threadIdx.x < sb && u < damp.n_free * 6) {
            const uint32_t f = u / 6, a = u % 6;
            const double d = damp.diagB[u];
            double s = damp.sc[u];
            if (damp.init_scale) {
                s = damp.jacobi ? 1.0 / (1.0 + sqrt(d)) : 1.0;
                damp.sc[u] = s;
            }
            const double lam = point_lambda(d, s, damp.lm);
            damp.lc[u] = lam;
            blocks[36 * (size_t)damp.diag_block[f] + a * 7] += lam;
        }
        __syncthreads();
    }
    const size_t sb2 = (size_t)sb * sb;
    double* dst = (is_f ? v.F : v.D) + i * sb2;
    const int32_t* mp = map + ((size_t)(is_f ? v.m : 0) + i) * bw * bw;
    // lane = column, wave = every fourth row: divisions by the constant 6 only, coalesced stores, independent loads
    const int c = threadIdx.x % 64, lb = c / 6, cc = c - lb * 6;
    if (c >= sb) return;
    double extra = 0.0;
    // blocks of S outside the block-tridiagonal part are not factored (off-band mode).  The band part of a positive definite
    // matrix need not be positive definite; a share comp_scale of the left-out blocks' absolute row sums on the diagonal makes
    // it so (at comp_scale = 1 the factored matrix exceeds S by a diagonally dominant one) - the caller raises the share only
    // when a factorisation breaks down, because what is added here weakens the preconditioner on the smooth modes
    if (comp && !is_f) {
        const uint32_t u = i * (uint32_t)sb + (uint32_t)c;
        if (u < v.n_real) extra += comp_scale * comp[u];
    }
#pragma unroll 4
    for (int r = threadIdx.x / 64; r < sb; r += 4) {
        const int la = r / 6, rr = r - la * 6;
        double val = 0.0;
        if (is_f) {
            const int32_t id = mp[la * bw + lb];
            if (id >= 0) val = blocks[36 * (size_t)id + cc * 6 + rr];
        } else {
            const int32_t id = la <= lb ? mp[la * bw + lb] : mp[lb * bw + la];
            if (id >= 0) val = la <= lb ? blocks[36 * (size_t)id + rr * 6 + cc] : blocks[36 * (size_t)id + cc * 6 + rr];
            else if (id == -2 && r == c) val = 1.0;
            if (r == c) val += extra;
        }
        dst[(size_t)r * sb + c] = val;
    }
}

// Eliminated node i = h (2 j + 1)  (or node 0 when final != 0): E_i = D_i^-1 by a block Gauss-Jordan sweep.
// Sweeping the 3x3 pivot block K of a matrix turns   A_KK -> A_KK^-1,  A_Kj -> A_KK^-1 A_Kj,  A_iK -> -A_iK A_KK^-1,
// A_ij -> A_ij - A_iK A_KK^-1 A_Kj;  after all 2 bw pivot blocks the array holds A^-1.
//
// The matrix (padded to 64 x 64) lives in the ACCUMULATOR registers of the f64 matrix cores: wave ti of four owns the
// tile row ti, four 16x16 tiles, D[i = 4 r + l/16][j = l%16] per tile.  A step is one rank-3 (padded to 4) update
// new = base + A' B on v_mfma_f64_16x16x4_f64 - four matrix instructions per wave instead of eighteen vector FMAs
// plus the cross-lane traffic that feeds them, and one wave per SIMD instead of nine on four:
//   * the waves that hold the pivot rows a = A_K: copy them to an LDS strip (the only exchange, one barrier per step);
//   * the sweep keeps the matrix symmetric up to the sign of the swept/unswept cross blocks, so the pivot COLUMNS need
//     no second exchange: A_iK = s_i a_i^T with s_i = -1 on rows swept earlier, +1 otherwise;
//   * every wave inverts the pivot block itself (closed form, redundant but off the other waves' critical path) and
//     forms its operands: A'[i][k] = -s_i a[k][i] (e_c on pivot row c), B[k][j] = (P^-1 a)[k][j] (P^-1[k][c] on pivot
//     column c); base = the tile with the pivot rows and columns zeroed.  Then pivot rows become P^-1 a, pivot columns
//     -A_iK P^-1, the pivot block P^-1, everything else A_ij - A_iK P^-1 A_Kj.
constexpr int kCrInvertThreads = 256;

// The sweep proper: the 64 x 64 image in the accumulator registers of four waves, in place.  Wave w owns tile COLUMN w:
// acc[tr][r] at lane (lk = l/16, lr = l%16) is the element (16 tr + 4 r + lk, 16 w + lr).  strips: [buffer 2][pivot row c,
// 3 = zeros][column 64] in LDS; pblk: [buffer 2][9 of 16] the pivot block itself.  Returns false when a pivot block was not
// positive definite.
//
// A step is a chain: exchange of the pivot rows -> inverse of the pivot block -> operands -> matrix instruction -> the NEXT
// pivot rows ready for their exchange, and a single wave pays every instruction of it in full (measured on gfx950, one wave
// per SIMD: an f64 vector instruction 13 cycles whether or not it depends on the one before, v_mfma_f64_16x16x4_f64 64, its
// result in a vector register 48 later, LDS store -> barrier -> load 156, readfirstlane round trip 60; scripts/probes/).  So
// the step is arranged to need few instructions on that chain:
//   * a wave owns a tile column.  Every wave then holds ONE tile of the next pivot rows: it updates that tile first, publishes
//     its sixteen columns of the rows (one store when the three rows share an accumulator register, else two) and leaves the
//     other three updates to the start of the next step, where they run beside that step's LDS requests.  The B operand is the
//     same for a wave's four tiles (strip[k][16 w + lr], no arithmetic); the A operand differs per tile row (three
//     multiply-adds each).  (One wave per tile ROW, the earlier form, made the owner of the next pivot rows wait for its four
//     matrix instructions and then write twelve row pieces under three lane masks while the other waves sat at the barrier:
//     1 230 cycles per step against 1 050.)
//   * the pivot rows are published with the IDENTITY in the pivot columns (the pivot block itself goes to pblk), and the
//     accumulators are zeroed on pivot rows (by the lanes that publish them) and pivot columns.  Then one formula covers
//     every position:  new = base + A' B,  A'[i][k] = -s_i sum_c strip[c][i] Pinv[c][k]  with s_i = -1 for rows up to and
//     including the pivot rows, +1 below: pivot rows become Pinv a, pivot columns -A_iK Pinv, the pivot block Pinv, the rest
//     A_ij - A_iK Pinv A_Kj;
//   * lane group lk needs row lk of Pinv only.  It reads the pivot block through the cyclic permutation that puts row lk
//     first and evaluates ONE row of the adjugate (three cofactors, no selects); the determinant is taken from lane 0 for
//     everybody.  A symmetric pair of entries comes out of the same two products rounded the same way in both groups, so
//     the inverse is bitwise symmetric - the sweep relies on it (rows taken from differently rounded inverses were
//     measured to cost four digits of the factor).
// NT: tile rows = tile columns = waves of the workgroup (4: the 64 x 64 image).  NT = 8 - a 128 x 128 image on eight waves - was built
// and measured as a one-sweep replacement of dense2_solve's two-block elimination at 19 cameras: 51 us against 43.  Two waves share
// each SIMD's f64 and matrix pipes, a step carries four times the matrix work of a 64-row step (eight tiles per wave, two waves), and
// 38 such steps are more than 2 x 20 small ones plus the two block products.  The template parameter stays for what it is worth to a
// cyclic reduction with super-blocks wider than ten cameras.
template <int NSTEPS, int NT = 4>
__device__ __forceinline__ bool gj_sweep(double4_t (&acc)[NT], double (*strips)[4][16 * NT], double (*pblk)[16], const int wave, const int lane)
{
    constexpr int W = 16 * NT;                          // image width = strip length
    static_assert(3 * NSTEPS <= W, "pivot rows inside the image");
    const int lr = lane % 16, lk = lane / 16;
    const int g = lk < 3 ? lk : 0;                       // the padding group computes group 0's row and multiplies zeros with it
    const int p0 = g, p1 = (g + 1) % 3, p2 = (g + 2) % 3;
    // the pivot block is read from its upper triangle only (the accumulators are symmetric up to rounding; every group must
    // see the same numbers)
    const int u01 = (p0 < p1 ? p0 : p1) * 3 + (p0 < p1 ? p1 : p0), u02 = (p0 < p2 ? p0 : p2) * 3 + (p0 < p2 ? p2 : p0),
              u12 = (p1 < p2 ? p1 : p2) * 3 + (p1 < p2 ? p2 : p1);
    // strip rows this lane reads: its B operand row (zeros for the padding k = 3) and the three pivot rows in permuted order
    const int zb = (lk < 3 ? lk : 3) * W + 16 * wave + lr;
    const int za0 = (lk < 3 ? p0 : 3) * W + lr, za1 = (lk < 3 ? p1 : 3) * W + lr, za2 = (lk < 3 ? p2 : 3) * W + lr;
    const int jcol = 16 * wave + lr;                     // the column this lane holds
    if (threadIdx.x < 2 * W) strips[threadIdx.x / W][3][threadIdx.x % W] = 0.0;   // (64 NT lanes in the workgroup: >= 2 W)
    bool ok = true;
    int bad = 0;   // NT = 8: see the pivot test below
    // publish the pivot rows of step kb (static), then zero them.  Rows that sit in the same accumulator register (two steps
    // in four all three do) go out with one store: lane group lk holds row k0 + (lk - k0 % 4).
    auto publish = [&](const int kb) __attribute__((always_inline)) {
        const int k0 = 3 * kb;
        double* strip = &strips[kb & 1][0][0];
        double* pb = &pblk[kb & 1][0];
        const int cj = jcol - k0;                        // 0..2: this lane holds a pivot column
        // group A: the rows in the register of row k0 (lane groups k0 % 4 ..), group B: the rest, in the next register
        const int qa = k0 / 4, qb = (k0 + 2) / 4, nA = qa == qb ? 3 : 4 - k0 % 4;
        {
            const int c = lk - k0 % 4;
            if (c >= 0 && c < nA) {
                double val = acc[qa / 4][qa % 4];
                if (cj >= 0 && cj < 3) {
                    pb[c * 3 + cj] = val;
                    val = cj == c ? 1.0 : 0.0;
                }
                strip[c * W + jcol] = val;
                acc[qa / 4][qa % 4] = 0.0;
            }
        }
        if (qb != qa) {
            const int c = nA + lk;
            if (c < 3) {
                double val = acc[qb / 4][qb % 4];
                if (cj >= 0 && cj < 3) {
                    pb[c * 3 + cj] = val;
                    val = cj == c ? 1.0 : 0.0;
                }
                strip[c * W + jcol] = val;
                acc[qb / 4][qb % 4] = 0.0;
            }
        }
    };
    publish(0);
    double aop_t[NT], sv_t = 0.0;   // operands of the updates the previous step left for this one
#pragma unroll
    for (int tr = 0; tr < NT; tr++) aop_t[tr] = 0.0;
    //   // operands of the updates the previous step left for this one
#pragma unroll
    for (int kb = 0; kb < NSTEPS; kb++) {   // straight-line code: every register index below is static
        const int k0 = 3 * kb;
        const double* strip = &strips[kb & 1][0][0];
        const double* pb = &pblk[kb & 1][0];
        __syncthreads();
        // requests first, arithmetic behind them
        const double a = pb[p0 * 4], b = pb[u01], c = pb[u02], d = pb[p1 * 4], e = pb[u12], f = pb[p2 * 4];
        const bool more = kb + 1 < NSTEPS;
        const int n0 = more ? (k0 + 3) / 16 : -1, n1 = more ? (k0 + 5) / 16 : -1;   // static: the tile rows of the next pivot rows
        // the pivot rows' entries this lane's A operands need.  Eight tile rows (NT = 8): only those of the tiles updated before
        // the exchange are requested up front, the rest behind it (48 registers of requests in flight would spill)
        double ac[NT][3];
#pragma unroll
        for (int tr = 0; tr < NT; tr++)
            if (NT <= 4 || !more || tr == n0 || tr == n1) { ac[tr][0] = strip[za0 + 16 * tr]; ac[tr][1] = strip[za1 + 16 * tr]; ac[tr][2] = strip[za2 + 16 * tr]; }
        const double sv = strip[zb];
        // the previous step's updates of the tiles that did not hold this step's pivot rows: their matrix instructions run
        // while the requests above are under way
        if (kb > 0) {
            const int m0 = k0 / 16, m1 = (k0 + 2) / 16;   // the tile rows the previous step updated before its exchange
#pragma unroll
            for (int tr = 0; tr < NT; tr++)
                if (tr != m0 && tr != m1) acc[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop_t[tr], sv_t, acc[tr], 0, 0, 0);
        }
        // row 0 of the inverse of the permuted block = row lk of Pinv in the column order p0 p1 p2
        const double c00 = __builtin_fma(d, f, -(e * e)), c01 = __builtin_fma(c, e, -(b * f)), c02 = __builtin_fma(b, e, -(c * d));
        const double det_own = __builtin_fma(a, c00, __builtin_fma(b, c01, c * c02));
        const double det = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(det_own)),
                                            __builtin_amdgcn_readfirstlane(__double2loint(det_own)));
        if constexpr (NT <= 4) {
            ok = ok && (a > 0.0) && (c00 > 0.0) && (det > 0.0);
        } else {
            // the same test, pinned to its step: left to itself the compiler sinks all 3 NSTEPS comparisons behind the last step and
            // keeps their operands alive until then - 228 registers at 38 steps, the whole budget of a wave of eight
            const int fail = (a > 0.0) && (c00 > 0.0) && (det > 0.0) ? 0 : 1;
            asm volatile("v_or_b32 %0, %0, %1" : "+v"(bad) : "v"(fail));
        }
        double id = __builtin_amdgcn_rcp(det);
        id = id * (2.0 - det * id);
        id = id * (2.0 - det * id);
        const double r0 = c00 * id, r1 = c01 * id, r2 = c02 * id;
        const int cj = jcol - k0;                        // 0..2: this lane holds a pivot column: base zero
        const bool pcol = cj >= 0 && cj < 3;
        // A'[i][k = lk], i = 16 tr + lr
        auto operand = [&](const int tr) __attribute__((always_inline)) {
            const double v = ac[tr][0] * r0 + ac[tr][1] * r1 + ac[tr][2] * r2;
            return 16 * tr + lr < k0 + 3 ? v : -v;
        };
        // pivot columns start from zero: only the wave(s) that hold them (a scalar branch), the lanes under the execution mask
        if (wave == k0 / 16 || wave == (k0 + 2) / 16) {
            if (pcol) {
#pragma unroll
                for (int tr = 0; tr < NT; tr++)
#pragma unroll
                    for (int r = 0; r < 4; r++) asm volatile("v_mov_b64 %0, 0" : "=v"(acc[tr][r]));
            }
        }
        if (more) {
            // the tile row(s) of the next pivot rows now, their exchange behind them; the other tiles' operands are kept
            acc[n0] = __builtin_amdgcn_mfma_f64_16x16x4f64(operand(n0), sv, acc[n0], 0, 0, 0);
            if (n1 != n0) acc[n1] = __builtin_amdgcn_mfma_f64_16x16x4f64(operand(n1), sv, acc[n1], 0, 0, 0);
#pragma unroll
            for (int tr = 0; tr < NT; tr++)
                if (tr != n0 && tr != n1) {
                    if (NT > 4) { ac[tr][0] = strip[za0 + 16 * tr]; ac[tr][1] = strip[za1 + 16 * tr]; ac[tr][2] = strip[za2 + 16 * tr]; }
                    aop_t[tr] = operand(tr);
                }
            sv_t = sv;
            __builtin_amdgcn_sched_barrier(0);   // the operands above fill the wait for the matrix instruction's result
            publish(kb + 1);
        } else {
#pragma unroll
            for (int tr = 0; tr < NT; tr++) acc[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(operand(tr), sv, acc[tr], 0, 0, 0);
        }
    }
    return ok && bad == 0;
}


template <int NSTEPS>
__global__ __launch_bounds__(kCrInvertThreads) void cr_invert_kernel(const CrView v, const uint32_t h, const int final_node,
                                                                     double* __restrict__ scal)
{
    __shared__ double strips[2][4][64];   // [buffer][pivot row c, 3 = zeros][column]
    __shared__ double pblk[2][16];        // [buffer][the pivot block]
    __shared__ double X[64 * kLd];
    const int sb = v.sb;
    const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const int lr = lane % 16, lk = lane / 16;
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t node = final_node ? 0u : h * (2 * blockIdx.x + 1);
    if (node >= v.m) return;

    double4_t acc[4];
    {
        const double* src = v.D + node * sb2;
#pragma unroll
        for (int tr = 0; tr < 4; tr++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * tr + 4 * r + lk, j = 16 * wave + lr;      // wave w: tile column w (gj_sweep)
                acc[tr][r] = (i < sb && j < sb) ? src[(size_t)i * sb + j] : (i == j ? 1.0 : 0.0);   // identity padding: extra steps are harmless
            }
    }

    const bool ok = gj_sweep<NSTEPS>(acc, strips, pblk, wave, lane);
    if (!ok) scal[SC_LIN_STATUS] = 1.0;   // every lane tests its own part of the pivot blocks; all write the same word
    // symmetric to rounding; store the symmetrised inverse (coalesced, through an LDS image)
#pragma unroll
    for (int tr = 0; tr < 4; tr++)
#pragma unroll
        for (int r = 0; r < 4; r++) X[(16 * tr + 4 * r + lk) * kLd + 16 * wave + lr] = acc[tr][r];
    __syncthreads();
    double* dst = v.E + node * sb2;
    for (int i = wave; i < sb; i += kCrInvertThreads / 64)
        if (lane < sb) dst[(size_t)i * sb + lane] = 0.5 * (X[i * kLd + lane] + X[lane * kLd + i]);
}

// 2 bw pivot blocks, rounded up to the next instantiated step count (the identity padding makes extra steps no-ops)
void launch_cr_invert(hipStream_t s, uint32_t grid, const CrView& v, uint32_t h, int final_node, double* scal)
{
    static_assert(kCrBandMax == 10, "step counts below cover bw <= 10 (20 pivot blocks of a 64 x 64 register image)");
    if (v.bw <= 3) hipLaunchKernelGGL(cr_invert_kernel<6>, dim3(grid), dim3(kCrInvertThreads), 0, s, v, h, final_node, scal);
    else if (v.bw <= 6) hipLaunchKernelGGL(cr_invert_kernel<12>, dim3(grid), dim3(kCrInvertThreads), 0, s, v, h, final_node, scal);
    else if (v.bw <= 9) hipLaunchKernelGGL(cr_invert_kernel<18>, dim3(grid), dim3(kCrInvertThreads), 0, s, v, h, final_node, scal);
    else hipLaunchKernelGGL(cr_invert_kernel<20>, dim3(grid), dim3(kCrInvertThreads), 0, s, v, h, final_node, scal);
}

// Remaining node k = 2 h j, column tile q = blockIdx.y (16 columns of every product), one wave per 16-row tile.
// All five operand matrices are requested from memory before the first LDS write (one latency, not five).
constexpr int kStageRegs = 60 / (kCrReduceThreads / 64);   // rows per wave at sb = 60: 15

struct StageRegs { double v[kStageRegs]; };

// wave w requests rows w, w + 4, .. (lane = column): coalesced, no index arithmetic beyond adds
__device__ __forceinline__ void stage_load(StageRegs& r, const double* __restrict__ src, int sb, int wave, int lane, bool on)
{
#pragma unroll
    for (int u = 0; u < kStageRegs; u++) {
        const int row = wave + u * (kCrReduceThreads / 64);
        r.v[u] = (on && row < sb && lane < sb) ? src[(size_t)row * sb + lane] : 0.0;
    }
}

// registers -> LDS image with leading dimension kLd (the padding was zeroed beforehand); transposed on request
__device__ __forceinline__ void stage_store(const StageRegs& r, double* __restrict__ dst, int sb, int wave, int lane, bool transpose)
{
#pragma unroll
    for (int u = 0; u < kStageRegs; u++) {
        const int row = wave + u * (kCrReduceThreads / 64);
        if (row < sb && lane < sb) dst[transpose ? lane * kLd + row : row * kLd + lane] = r.v[u];
    }
}

// The fill F_k <- -Q_a^T F_a goes to Fout, NOT over F_k: the column-tile workgroups of a node all read the whole of the
// old F_k, and nothing orders one workgroup's stores against another's loads inside a launch (on a busy or shared device
// they start at different times - a workgroup arriving late would stage rows a sibling has already replaced).
//
// The forward sweep of a solve rides along (fwd.w != nullptr): a right-hand side that is known when the factorisation starts
// (the first PCG round's b) needs  w_k = src_k - Q_a^T src_a - P_c^T src_c  at this level, and the column tile q of Q_a and
// of P_c that the sum needs lies in LDS here anyway - the workgroup adds its sixteen entries of w_k (sixteen partial sums per
// entry, fixed order), in place: a level writes the nodes k and reads the nodes a, c, which are different nodes.  The solve
// then starts at the top of the tree (launch_cr_solve, forward_done): two launches fewer per solve.
struct CrFwd {
    const double* src;   // values of the nodes a, c, k before this level: the caller's right-hand side at h = 1, w afterwards
    uint32_t n_src;      // valid entries of src (it reads as zero beyond them)
    double* w;           // padded work vector (m * sb); nullptr: no forward sweep
};

__global__ __launch_bounds__(kCrReduceThreads) void cr_reduce_kernel(const CrView v, const uint32_t h, const double* __restrict__ Fin,
                                                                     double* __restrict__ Fout, const uint32_t stagger_10ns, const CrFwd fwd)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double fy[2][64];          // src_a, src_c
    __shared__ double fred[2][16][16];    // partial sums of Q_a^T src_a and P_c^T src_c: [a / c][part][entry]
    // Development hook (SOSLAM_CR_STAGGER): the first column tile of
    // every node starts late by the given time, as it may on a busy or shared device, so that any dependence of one
    // workgroup of this launch on the loads of another shows every time instead of once in a while.  Bounded wait.
    if (stagger_10ns && blockIdx.y == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while (__builtin_amdgcn_s_memrealtime() - t0 < stagger_10ns) __builtin_amdgcn_s_sleep(16);
        asm volatile("" ::: "memory");   // the operand loads below must not be hoisted above the wait
    }
    const int sb = v.sb, tid = threadIdx.x;
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t k = 2 * h * blockIdx.x;
    if (k >= v.m) return;
    const bool has_a = k >= h, has_c = k + h < v.m;
    if (!has_a && !has_c) return;
    const uint32_t a = has_a ? k - h : 0, c = has_c ? k + h : 0;
    const bool fill = has_a && a >= h;   // node a had a left neighbour a-h = k-2h: new coupling A(k, k-2h)
    constexpr int kp = kImgRows;            // rows of every LDS image (zero beyond sb)
    const int nt = (sb + 15) / 16;          // 16-wide tiles per dimension (<= 4)
    const int wave = tid / 64, lane = tid % 64;
    const int q0 = (int)blockIdx.y * 16;     // this workgroup's column tile
    const int w0 = wave * 16;                // this wave's row tile
    const bool active = wave < nt;
    double* B0 = lds;
    double* B1 = B0 + kp * kLd;
    double* B2 = B1 + kp * kLd;
    double* B3 = B2 + kp * kLd;

    StageRegs rEa, rFk, rFa, rEc, rFc;
    stage_load(rEa, v.E + a * sb2, sb, wave, lane, has_a);
    stage_load(rFk, Fin + k * sb2, sb, wave, lane, has_a);
    stage_load(rFa, Fin + a * sb2, sb, wave, lane, fill);
    stage_load(rEc, v.E + c * sb2, sb, wave, lane, has_c);
    stage_load(rFc, Fin + c * sb2, sb, wave, lane, has_c);
    double dk[4];   // this wave's tile of D_k
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = w0 + 4 * r + lane / 16, col = q0 + lane % 16;
        dk[r] = (active && row < sb && col < sb) ? v.D[k * sb2 + (size_t)row * sb + col] : 0.0;
    }
    const bool do_fwd = fwd.w != nullptr;
    double fsrc = 0.0, fk = 0.0;
    if (do_fwd) {
        if (tid < 128) {
            const bool on = tid < 64 ? has_a : has_c;
            const size_t idx = (size_t)(tid < 64 ? a : c) * sb + (size_t)(tid % 64);
            fsrc = (on && tid % 64 < sb && idx < fwd.n_src) ? fwd.src[idx] : 0.0;
        } else if (tid < 144) {
            const int t = q0 + tid - 128;
            const size_t idx = (size_t)k * sb + (size_t)t;
            fk = (t < sb && idx < fwd.n_src) ? fwd.src[idx] : 0.0;
        }
    }
    // 16-byte stores (zeroing only the padding - a quarter of the images - was measured slower: its index arithmetic costs more
    // than the stores it saves)
    static_assert((4 * kImgRows * kLd) % 2 == 0, "16-byte zero fill");
    for (int e = tid; e < 2 * kp * kLd; e += kCrReduceThreads) reinterpret_cast<double2*>(lds)[e] = make_double2(0.0, 0.0);
    if (do_fwd && tid < 128) fy[tid / 64][tid % 64] = fsrc;
    __syncthreads();
    // sixteen lanes per entry of the column tile: part p sums the rows p, p + 16, .. of the tile in B2 against y
    auto fwd_partial = [&](const double* img, const double* y, double (*red)[16]) {
        const int j = tid % 16, part = tid / 16;
        double sum = 0.0;
#pragma unroll
        for (int m = part; m < kp; m += 16) sum += img[m * kLd + q0 + j] * y[m];
        red[part][j] = sum;
    };

    // tile (rows i0.., cols j0..) of the accumulator to global row-major (+ optionally transposed) and to an LDS image
    auto put = [&](const double4_t acc, int i0, int j0, double* out, double* outT, double* img, double sign) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = i0 + 4 * r + lane / 16, col = j0 + lane % 16;
            const double val = sign * acc[r];
            if (img && row < kp) img[row * kLd + col] = val;
            if (row < sb && col < sb) {
                if (out) out[(size_t)row * sb + col] = val;
                if (outT) outT[(size_t)col * sb + row] = val;
            }
        }
    };
    double4_t dacc = {0.0, 0.0, 0.0, 0.0};   // this wave's tile (w0, q0) of the update of D_k
    if (has_a) {
        stage_store(rEa, B0, sb, wave, lane, false);
        stage_store(rFk, B1, sb, wave, lane, true);             // B1[m][i] = F_k[i][m]
        if (fill) stage_store(rFa, B3, sb, wave, lane, false);
        __syncthreads();
        if (active) put(mfma_ptq_tile(B0, B1, w0, q0, lane), w0, q0, v.Q + a * sb2, v.QT + a * sb2, B2, 1.0);   // Q_a[:, q]
        __syncthreads();
        if (do_fwd) fwd_partial(B2, fy[0], fred[0]);
        if (active) {
            dacc = mfma_ptq_tile(B1, B2, w0, q0, lane);                                   // (F_k Q_a)[w, q]
            // rows q of the fill: -(Q_a[:, q])^T F_a, this wave takes column tile w
            if (fill) put(mfma_ptq_tile(B2, B3, q0, w0, lane), q0, w0, Fout + k * sb2, nullptr, nullptr, -1.0);
        }
        __syncthreads();
    }
    if (has_c) {
        stage_store(rEc, B0, sb, wave, lane, false);
        stage_store(rFc, B1, sb, wave, lane, false);
        __syncthreads();
        if (active) put(mfma_ptq_tile(B0, B1, w0, q0, lane), w0, q0, v.P + c * sb2, v.PT + c * sb2, B2, 1.0);   // P_c[:, q]
        __syncthreads();
        if (do_fwd) fwd_partial(B2, fy[1], fred[1]);
        if (active) {
            const double4_t s = mfma_ptq_tile(B1, B2, w0, q0, lane);                      // (F_c^T P_c)[w, q]
#pragma unroll
            for (int r = 0; r < 4; r++) dacc[r] += s[r];
        }
    }
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = w0 + 4 * r + lane / 16, col = q0 + lane % 16;
            if (row < sb && col < sb) v.D[k * sb2 + (size_t)row * sb + col] = dk[r] - dacc[r];
        }
    }
    if (do_fwd) {
        __syncthreads();
        if (tid >= 128 && tid < 144) {
            const int j = tid - 128, t = q0 + j;
            double sa = 0.0, sc = 0.0;
            if (has_a) {
#pragma unroll
                for (int p = 0; p < 16; p++) sa += fred[0][p][j];
            }
            if (has_c) {
#pragma unroll
                for (int p = 0; p < 16; p++) sc += fred[1][p][j];
            }
            if (t < sb) fwd.w[(size_t)k * sb + t] = fk - (sa + sc);
        }
    }
}


// ---- small dense systems: two-block elimination by ONE workgroup on the matrix cores ---------------------------------------
// The reference's sliding windows (slam.cpp:121-129) leave a dense reduced system of at most 19 free cameras (114 unknowns).
// S = [A B; B^T C] with A the first ten cameras: E_A = A^-1 (register-resident block Gauss-Jordan sweep, as cr_invert),
// T = E_A B, S_C = C - B^T T (both on the f64 matrix cores, operands as zero-padded LDS images), E_C = S_C^-1 (second sweep),
// then every solve is four 64 x 64 matrix-vector products:  y = E_A b_A,  z = b_C - B^T y,  x_C = E_C z,  x_A = y - T x_C.
// One round of iterative refinement against the block-sparse S itself (fixed-order row gather) takes the explicit inverses'
// rounding out of the result.  40 dependent pivot steps instead of the 19 x 3 barrier-separated phases of the blocked Cholesky
// (dense_small_solve, which stays for 21 and 22 cameras).
constexpr int kDense2Threads = 256;
constexpr int kDense2Cams = 10;      // cameras of the leading block (60 rows of the 64 x 64 image)
constexpr int kDense2Blocks = 2 * kDense2Cams * (2 * kDense2Cams + 1) / 2;   // upper blocks of a 20-camera system: 210

__global__ __launch_bounds__(kDense2Threads) void dense2_solve_kernel(const BsrView A, const uint32_t n_blocks,
                                                                      const uint32_t* __restrict__ blk_row,
                                                                      const uint32_t* __restrict__ blk_col,
                                                                      const double* __restrict__ b, double* __restrict__ x,
                                                                      double* __restrict__ scal)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double strips[2][4][64];
    __shared__ double pblk[2][16];
    __shared__ double vb[2][64], vx[2][64], vr[2][64], vy[64], vz[64], vc[64];
    __shared__ uint16_t brc[2 * kDense2Blocks];   // block coordinates (row, column)
    double* const IA = lds;                 // A, then E_A
    double* const IB = IA + 64 * kLd;       // B: rows of A's cameras, columns of C's
    double* const IT = IB + 64 * kLd;       // T = E_A B
    double* const IC = IT + 64 * kLd;       // C, then E_C = (C - B^T T)^-1
    const int tid = threadIdx.x, lane = tid % 64, wave = __builtin_amdgcn_readfirstlane(tid / 64);
    const int lr = lane % 16, lk = lane / 16;
    const int nb = (int)A.n_rows, nA = nb < kDense2Cams ? nb : kDense2Cams, nC = nb - nA, sA = 6 * nA, sC = 6 * nC;

    for (int e = tid; e < 2 * 64 * kLd; e += kDense2Threads) reinterpret_cast<double2*>(lds)[e] = make_double2(0.0, 0.0);
    __syncthreads();
    // the block coordinates go to LDS first (coalesced), then the values stream in (coalesced, eight loads in flight per lane)
    // and are scattered to the images
    for (uint32_t i = tid; i < n_blocks; i += kDense2Threads) { brc[i] = (uint16_t)blk_row[i]; brc[kDense2Blocks + i] = (uint16_t)blk_col[i]; }
    __syncthreads();
    for (uint32_t e0 = 0; e0 < n_blocks * 36; e0 += 8 * kDense2Threads) {
        double val8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t e = e0 + (uint32_t)u * kDense2Threads + (uint32_t)tid;
            val8[u] = e < n_blocks * 36 ? A.blocks[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t e = e0 + (uint32_t)u * kDense2Threads + (uint32_t)tid;
            if (e >= n_blocks * 36) continue;
            const uint32_t blk = e / 36, t = e - blk * 36;
            const int r = (int)t / 6, c = (int)t % 6;
            const int ca = (int)brc[blk], cb = (int)brc[kDense2Blocks + blk];   // ca <= cb
            const double val = val8[u];
            if (cb < nA) {
                IA[(6 * ca + r) * kLd + 6 * cb + c] = val;
                if (ca != cb) IA[(6 * cb + c) * kLd + 6 * ca + r] = val;
            } else if (ca >= nA) {
                IC[(6 * (ca - nA) + r) * kLd + 6 * (cb - nA) + c] = val;
                if (ca != cb) IC[(6 * (cb - nA) + c) * kLd + 6 * (ca - nA) + r] = val;
            } else {
                IB[(6 * ca + r) * kLd + 6 * (cb - nA) + c] = val;
            }
        }
    }
    if (tid < 64) {
        if (tid >= sA) IA[tid * kLd + tid] = 1.0;   // identity padding: the extra pivot steps are harmless
        if (tid >= sC) IC[tid * kLd + tid] = 1.0;
        vb[0][tid] = tid < sA ? b[tid] : 0.0;
        vb[1][tid] = tid < sC ? b[sA + tid] : 0.0;
    }
    __syncthreads();

    double4_t acc[4];
    auto load_acc = [&](const double* img) __attribute__((always_inline)) {
#pragma unroll
        for (int tr = 0; tr < 4; tr++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[tr][r] = img[(16 * tr + 4 * r + lk) * kLd + 16 * wave + lr];   // wave w: tile column w
    };
    auto store_acc = [&](double* img) __attribute__((always_inline)) {
#pragma unroll
        for (int tr = 0; tr < 4; tr++)
#pragma unroll
            for (int r = 0; r < 4; r++) img[(16 * tr + 4 * r + lk) * kLd + 16 * wave + lr] = acc[tr][r];
    };
    load_acc(IA);
    bool ok = gj_sweep<2 * kDense2Cams>(acc, strips, pblk, wave, lane);
    store_acc(IA);                          // nobody reads IA between the load above and here
    __syncthreads();
    if (nC > 0) {
#pragma unroll
        for (int tj = 0; tj < 4; tj++) {    // T = E_A B (E_A symmetric: P^T Q with P = E_A)
            const double4_t t4 = mfma_ptq_tile(IA, IB, 16 * wave, 16 * tj, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) IT[(16 * wave + 4 * r + lk) * kLd + 16 * tj + lr] = t4[r];
        }
        __syncthreads();
#pragma unroll
        for (int tr = 0; tr < 4; tr++) {    // S_C = C - B^T T, straight into the sweep's layout (wave w: tile column w)
            const double4_t p4 = mfma_ptq_tile(IB, IT, 16 * tr, 16 * wave, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) acc[tr][r] = IC[(16 * tr + 4 * r + lk) * kLd + 16 * wave + lr] - p4[r];
        }
        __syncthreads();                    // gj_sweep rewrites the strips' zero rows: everyone is past the first sweep's reads
        ok = gj_sweep<2 * kDense2Cams>(acc, strips, pblk, wave, lane) && ok;
        store_acc(IC);
        __syncthreads();
    }
    if (!ok) scal[SC_LIN_STATUS] = 1.0;   // as in cr_invert: every lane tests its own part of the pivot blocks

    // (dA, dC) = S^-1 (rA, rC) through the stored operators; four lanes per row (16 terms each, then a quad sum), fixed order
    const int row = tid >> 2, m0 = (tid & 3) * 16;
    auto solve = [&](const double* rA, const double* rC, double* dA, double* dC) __attribute__((always_inline)) {
        {
            double y = 0.0;
#pragma unroll
            for (int m = 0; m < 16; m++) y += IA[row * kLd + m0 + m] * rA[m0 + m];
            y = quad_sum(y);
            if ((tid & 3) == 0) vy[row] = y;
        }
        __syncthreads();
        if (nC > 0) {
            {
                double z = 0.0;
#pragma unroll
                for (int m = 0; m < 16; m++) z += IB[(m0 + m) * kLd + row] * vy[m0 + m];
                z = quad_sum(z);
                if ((tid & 3) == 0) vz[row] = rC[row] - z;
            }
            __syncthreads();
            {
                double c = 0.0;
#pragma unroll
                for (int m = 0; m < 16; m++) c += IC[row * kLd + m0 + m] * vz[m0 + m];
                c = quad_sum(c);
                if ((tid & 3) == 0) vc[row] = c;
            }
            __syncthreads();
            {
                double a = 0.0;
#pragma unroll
                for (int m = 0; m < 16; m++) a += IT[row * kLd + m0 + m] * vc[m0 + m];
                a = quad_sum(a);
                if ((tid & 3) == 0) { dA[row] = vy[row] - a; dC[row] = vc[row]; }
            }
        } else if ((tid & 3) == 0) {
            dA[row] = vy[row];
            dC[row] = 0.0;
        }
        __syncthreads();
    };
    solve(vb[0], vb[1], vx[0], vx[1]);
    // refinement: r = b - S x from the block-sparse matrix (the adjacency of the PCG kernels: every block row lists its blocks)
    {
        // two lanes per row share its blocks (every second one each), added in a fixed order
        const int rw = tid >> 1, half = tid & 1;
        const bool live = rw < 6 * nb;
        const int f = live ? rw / 6 : 0, a = rw % 6;
        double s = 0.0;
        if (live)
        for (uint32_t e = A.row_ptr[f] + (uint32_t)half; e < A.row_ptr[f + 1]; e += 2) {
            const double* B = A.blocks + 36 * (size_t)A.ent_blk[e];
            const int col = (int)A.ent_col[e];
            const double* xv = col < nA ? vx[0] + 6 * col : vx[1] + 6 * (col - nA);
            if (A.ent_trans[e]) {
#pragma unroll
                for (int c = 0; c < 6; c++) s -= B[c * 6 + a] * xv[c];
            } else {
#pragma unroll
                for (int c = 0; c < 6; c++) s -= B[a * 6 + c] * xv[c];
            }
        }
        s += mov_dpp_f64<kDppXor1>(s);
        if (live && half == 0) {
            s += b[rw];
            if (f < nA) vr[0][rw] = s; else vr[1][rw - sA] = s;
        }
    }
    if (tid < 64) {
        if (tid >= sA) vr[0][tid] = 0.0;
        if (tid >= sC) vr[1][tid] = 0.0;
    }
    __syncthreads();
    solve(vr[0], vr[1], vb[0], vb[1]);      // the correction lands in vb (the right-hand side is no longer needed)
    if (tid < 6 * nb) x[tid] = tid < sA ? vx[0][tid] + vb[0][tid] : vx[1][tid - sA] + vb[1][tid - sA];
}


// ---- dense SPD inverse, in place, by block Gauss-Jordan over 60 x 60 tiles ----------------------------------------------------
// The coarse matrix of the pose graph's two-level preconditioner (pg_solver.hip: ten aggregates of six rigid-body modes per
// tile, a few hundred to a thousand unknowns) is inverted explicitly, so that every PCG iteration applies it as one small dense
// product.  A is nb x nb tiles of 60 x 60, row-major with leading dimension 60 nb, symmetric positive definite.  Per pivot
// tile K:  E = A_KK^-1 (the register-resident sweep of cr_invert);  row: A_Kj <- E A_Kj;  update: A_ij -= A_iK A_Kj (new row,
// old column);  column: A_iK <- -A_iK E, A_KK <- E.  The column step of pivot K shares a launch with the inversion of pivot
// K + 1 (they touch different tiles): 3 nb + 1 launches, every one a grid of 64^3 products on the f64 matrix cores.
constexpr int kTile = 60;

__device__ __forceinline__ void tile_to_image(const double* __restrict__ src, int ld, double* __restrict__ img, bool transpose)
{
    // 256 lanes: wave w takes rows w, w + 4, ..; lane = column (coalesced)
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    for (int row = wave; row < kTile; row += 4)
        if (lane < kTile) {
            const double v = src[(size_t)row * ld + lane];
            img[transpose ? lane * kLd + row : row * kLd + lane] = v;
        }
}

// C (60 x 60 in global memory) = alpha * (P^T Q) + beta * C for LDS images P, Q (64 rows, zero beyond 60)
__device__ __forceinline__ void tile_product_store(const double* __restrict__ P, const double* __restrict__ Q, double* __restrict__ C, int ld,
                                                   double alpha, bool accumulate)
{
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
#pragma unroll
    for (int tj = 0; tj < 4; tj++) {
        const double4_t t = mfma_ptq_tile(P, Q, 16 * wave, 16 * tj, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wave + 4 * r + lane / 16, col = 16 * tj + lane % 16;
            if (row < kTile && col < kTile) {
                double* dst = C + (size_t)row * ld + col;
                *dst = accumulate ? *dst + alpha * t[r] : alpha * t[r];
            }
        }
    }
}

// blockIdx 0: E = A_KK^-1 -> Ebuf (pivot tile K); blockIdx 1..: column step of the PREVIOUS pivot Kp (tile rows i != Kp):
// A_iKp <- -A_iKp Eprev; the last of them also stores Eprev into A_KpKp.  Kp < 0: no previous pivot; K >= nb: no pivot left.
__global__ __launch_bounds__(256) void bgj_pivot_kernel(double* __restrict__ A, const int nb, const int K, const int Kp,
                                                        double* __restrict__ Ebuf, const double* __restrict__ Eprev, double* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double strips[2][4][64];
    __shared__ double pblk[2][16];
    const int ld = kTile * nb;
    const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const int lr = lane % 16, lk = lane / 16;
    if (blockIdx.x == 0) {
        if (K >= nb) return;
        const double* src = A + ((size_t)K * kTile) * ld + (size_t)K * kTile;
        double4_t acc[4];
#pragma unroll
        for (int tr = 0; tr < 4; tr++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * tr + 4 * r + lk, j = 16 * wave + lr;
                acc[tr][r] = (i < kTile && j < kTile) ? src[(size_t)i * ld + j] : (i == j ? 1.0 : 0.0);
            }
        const bool ok = gj_sweep<kTile / 3>(acc, strips, pblk, wave, lane);
        if (!ok) status[0] = 1.0;
        double* X = lds;
#pragma unroll
        for (int tr = 0; tr < 4; tr++)
#pragma unroll
            for (int r = 0; r < 4; r++) X[(16 * tr + 4 * r + lk) * kLd + 16 * wave + lr] = acc[tr][r];
        __syncthreads();
        for (int i = wave; i < kTile; i += 4)
            if (lane < kTile) Ebuf[i * kTile + lane] = 0.5 * (X[i * kLd + lane] + X[lane * kLd + i]);
        return;
    }
    if (Kp < 0) return;
    int i = (int)blockIdx.x - 1;
    if (i >= Kp) i++;                      // tile rows 0 .. nb - 1 without Kp
    double* B0 = lds;                      // A_iKp^T
    double* B1 = B0 + kImgRows * kLd;      // Eprev
    for (int e = threadIdx.x; e < kImgRows * kLd; e += 256) reinterpret_cast<double2*>(lds)[e] = make_double2(0.0, 0.0);
    __syncthreads();
    if (i < nb) {
        double* tile = A + ((size_t)i * kTile) * ld + (size_t)Kp * kTile;
        tile_to_image(tile, ld, B0, true);
        tile_to_image(Eprev, kTile, B1, false);
        __syncthreads();
        tile_product_store(B0, B1, tile, ld, -1.0, false);
    } else {
        // one more workgroup: the pivot tile itself becomes Eprev
        double* tile = A + ((size_t)Kp * kTile) * ld + (size_t)Kp * kTile;
        for (int row = wave; row < kTile; row += 4)
            if (lane < kTile) tile[(size_t)row * ld + lane] = Eprev[row * kTile + lane];
    }
}

// row step: A_Kj <- E A_Kj for j != K
__global__ __launch_bounds__(256) void bgj_row_kernel(double* __restrict__ A, const int nb, const int K, const double* __restrict__ E)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ld = kTile * nb;
    int j = (int)blockIdx.x;
    if (j >= K) j++;
    double* B0 = lds;                      // E (symmetric: P^T = E)
    double* B1 = B0 + kImgRows * kLd;      // A_Kj
    for (int e = threadIdx.x; e < kImgRows * kLd; e += 256) reinterpret_cast<double2*>(lds)[e] = make_double2(0.0, 0.0);
    __syncthreads();
    double* tile = A + ((size_t)K * kTile) * ld + (size_t)j * kTile;
    tile_to_image(E, kTile, B0, false);
    tile_to_image(tile, ld, B1, false);
    __syncthreads();
    tile_product_store(B0, B1, tile, ld, 1.0, false);
}

// update: A_ij -= A_iK A_Kj for i != K, j != K (A_Kj: the new row, A_iK: the old column)
__global__ __launch_bounds__(256) void bgj_update_kernel(double* __restrict__ A, const int nb, const int K)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ld = kTile * nb;
    int i = (int)blockIdx.y, j = (int)blockIdx.x;
    if (i >= K) i++;
    if (j >= K) j++;
    double* B0 = lds;                      // A_iK^T
    double* B1 = B0 + kImgRows * kLd;      // A_Kj
    for (int e = threadIdx.x; e < kImgRows * kLd; e += 256) reinterpret_cast<double2*>(lds)[e] = make_double2(0.0, 0.0);
    __syncthreads();
    tile_to_image(A + ((size_t)i * kTile) * ld + (size_t)K * kTile, ld, B0, true);
    tile_to_image(A + ((size_t)K * kTile) * ld + (size_t)j * kTile, ld, B1, false);
    __syncthreads();
    tile_product_store(B0, B1, A + ((size_t)i * kTile) * ld + (size_t)j * kTile, ld, -1.0, true);
}

// ---- solve phase: every step is a matrix-vector product with a stored sb x sb operator ------------------------
// w holds the right-hand side on entry (padded to m*sb), the solution on exit.

__device__ __forceinline__ bool cr_done(const double* __restrict__ done_flag) { return done_flag && *done_flag != 0.0; }

// The solve kernels use 512 lanes per node: wave g of 8 sums every 8th product of output element t = lane (operator
// columns: consecutive lanes read consecutive doubles; the vector element is wave-uniform, a scalar load), all loads
// of a lane are issued before its first multiply, and the eight partial sums are added in a fixed order through
// LDS.  These kernels are pure latency: one memory round trip each.
constexpr int kCrSolveThreads = 512;
constexpr int kCrSolveGroups = kCrSolveThreads / 64;
constexpr int kCrSolveIters = (60 + kCrSolveGroups - 1) / kCrSolveGroups;   // 8 products per lane and operator at sb = 60

__device__ __forceinline__ double group_sum(double part, double (*red)[64], int t, int g)
{
    red[g][t] = part;
    __syncthreads();
    double s = red[0][t];
#pragma unroll
    for (int q = 1; q < kCrSolveGroups; q++) s += red[q][t];
    return s;
}

struct ColOp { double m[kCrSolveIters], y[kCrSolveIters]; };

// request M[m][t] and y[y0 + m] for m = g, g + 8, ..; the vector has n_y valid entries and reads as zero beyond them
// (the right-hand side of a solve is the caller's unpadded vector on first touch, the padded work vector afterwards)
__device__ __forceinline__ void col_load(ColOp& o, const double* __restrict__ M, const double* __restrict__ y, size_t y0, uint32_t n_y,
                                         int sb, int t, int g, bool on)
{
#pragma unroll
    for (int u = 0; u < kCrSolveIters; u++) {
        const int m = g + u * kCrSolveGroups;
        const bool in = on && m < sb && t < sb;
        o.m[u] = in ? M[(size_t)m * sb + t] : 0.0;
        o.y[u] = (in && y0 + m < n_y) ? y[y0 + m] : 0.0;
    }
}

__device__ __forceinline__ double col_dot(const ColOp& o)
{
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < kCrSolveIters; u++) s += o.m[u] * o.y[u];
    return s;
}

// forward, remaining node k: w_k = src_k - Q_a^T src_a - P_c^T src_c  (src = the caller's right-hand side at h = 1: no copy
// into the work vector; = w afterwards)
__global__ __launch_bounds__(kCrSolveThreads) void cr_fwd_kernel(const CrView v, const uint32_t h, const double* __restrict__ src,
                                                                 const uint32_t n_src, double* __restrict__ w,
                                                                 const double* __restrict__ done_flag)
{
    __shared__ double red[kCrSolveGroups][64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t k = 2 * h * blockIdx.x;
    if (k >= v.m) return;
    const bool has_a = k >= h, has_c = k + h < v.m;
    const uint32_t a = has_a ? k - h : 0, c = has_c ? k + h : 0;
    ColOp oa, oc;
    col_load(oa, v.Q + a * sb2, src, (size_t)a * sb, n_src, sb, t, g, has_a);
    col_load(oc, v.P + c * sb2, src, (size_t)c * sb, n_src, sb, t, g, has_c);
    const size_t ik = (size_t)k * sb + t;
    const double wk = (g == 0 && t < sb && ik < n_src) ? src[ik] : 0.0;
    const double s = group_sum(col_dot(oa) + col_dot(oc), red, t, g);
    if (g == 0 && t < sb) w[ik] = wk - s;
}

// backward, node i eliminated at stride h: x_i = E_i own_i - P_i x_{i-h} - Q_i x_{i+h}; own = the caller's right-hand side
// at h = 1 (those nodes were never touched by the forward sweep), w otherwise.  The result goes to the padded solution
// vector xs (later levels read their neighbours there; w keeps the right-hand sides, which fused levels read again) and
// straight to the caller's x.
__global__ __launch_bounds__(kCrSolveThreads) void cr_bwd_kernel(const CrView v, const uint32_t h, const double* __restrict__ own,
                                                                 const uint32_t n_own, double* __restrict__ xs, double* __restrict__ x,
                                                                 const uint32_t n_x, const double* __restrict__ done_flag)
{
    __shared__ double red[kCrSolveGroups][64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t i = h * (2 * blockIdx.x + 1);
    if (i >= v.m) return;
    const bool has_c = i + h < v.m;
    const uint32_t a = i - h, c = has_c ? i + h : 0;
    const uint32_t n_w = v.m * (uint32_t)sb;
    ColOp oe, oa, oc;
    col_load(oe, v.E + i * sb2, own, (size_t)i * sb, n_own, sb, t, g, true);   // E symmetric
    col_load(oa, v.PT + i * sb2, xs, (size_t)a * sb, n_w, sb, t, g, true);
    col_load(oc, v.QT + i * sb2, xs, (size_t)c * sb, n_w, sb, t, g, has_c);
    const double s = group_sum(col_dot(oe) - col_dot(oa) - col_dot(oc), red, t, g);
    if (g == 0 && t < sb) {
        const size_t ii = (size_t)i * sb + t;
        xs[ii] = s;
        if (ii < n_x) x[ii] = s;
    }
}

// ---- two levels of a sweep in one launch --------------------------------------------------------------------------
// A launch costs about as much as the memory round trip it contains, so two consecutive levels are fused by
// recomputing what a workgroup needs from the lower level itself: the forward pair (h, 2h) has one workgroup per node
// k = 4h j, which forms the level-h values of k - 2h, k and k + 2h (six products, three of them also formed by a
// neighbour) and then its own level-2h value; the backward pair (2h, h) has one workgroup per level-h node, which first
// solves its level-2h neighbour.  Every value is written by exactly one workgroup, and the arithmetic per node and its
// order are those of cr_fwd_kernel / cr_bwd_kernel (the results are bitwise the same).
struct MCol { double m[kCrSolveIters]; };

__device__ __forceinline__ void mcol_load(MCol& o, const double* __restrict__ M, int sb, int t, int g, bool on)
{
#pragma unroll
    for (int u = 0; u < kCrSolveIters; u++) {
        const int m = g + u * kCrSolveGroups;
        o.m[u] = (on && m < sb && t < sb) ? M[(size_t)m * sb + t] : 0.0;
    }
}

// sum_m M[m][t] y[m], m = g, g + 8, ..; y: 64 doubles in LDS, zero from sb on
__device__ __forceinline__ double mcol_dot(const MCol& o, const double* __restrict__ y, int g)
{
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < kCrSolveIters; u++) {
        const int m = g + u * kCrSolveGroups;
        s += o.m[u] * (m < 64 ? y[m] : 0.0);
    }
    return s;
}

__global__ __launch_bounds__(kCrSolveThreads) void cr_fwd2_kernel(const CrView v, const uint32_t h, const double* __restrict__ src,
                                                                  const uint32_t n_src, double* __restrict__ w,
                                                                  const double* __restrict__ done_flag)
{
    __shared__ double red[3][kCrSolveGroups][64];
    __shared__ double yv[7][64];    // src of the nodes k - 3h .. k + 3h
    __shared__ double w1[3][64];    // level-h values of k - 2h, k, k + 2h
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const size_t sb2 = (size_t)sb * sb;
    const int64_t m = v.m, hh = h, k = 4 * hh * blockIdx.x;
    if (k >= m) return;
    const bool has[3] = {k >= 2 * hh, true, k + 2 * hh < m};
    MCol qa[3], pc[3], q2, p2;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int64_t tn = k + (j - 1) * 2 * hh;
        const bool ha = has[j] && tn >= hh, hc = has[j] && tn + hh < m;
        mcol_load(qa[j], v.Q + (ha ? tn - hh : 0) * sb2, sb, t, g, ha);
        mcol_load(pc[j], v.P + (hc ? tn + hh : 0) * sb2, sb, t, g, hc);
    }
    mcol_load(q2, v.Q + (has[0] ? k - 2 * hh : 0) * sb2, sb, t, g, has[0]);
    mcol_load(p2, v.P + (has[2] ? k + 2 * hh : 0) * sb2, sb, t, g, has[2]);
    if (g < 7) {
        const int64_t node = k + (g - 3) * hh;
        const size_t idx = (size_t)(node >= 0 ? node : 0) * sb + t;
        yv[g][t] = (node >= 0 && node < m && t < sb && idx < n_src) ? src[idx] : 0.0;
    }
    __syncthreads();
    // level h: node k + (j - 1) 2h reads its neighbours' vectors 2 j and 2 j + 2, its own is 2 j + 1
#pragma unroll
    for (int j = 0; j < 3; j++) red[j][g][t] = mcol_dot(qa[j], yv[2 * j], g) + mcol_dot(pc[j], yv[2 * j + 2], g);
    __syncthreads();
    if (g < 3) {
        double s = red[g][0][t];
#pragma unroll
        for (int q = 1; q < kCrSolveGroups; q++) s += red[g][q][t];
        const double val = has[g] ? yv[2 * g + 1][t] - s : 0.0;
        w1[g][t] = t < sb ? val : 0.0;
        if (g == 2 && has[2] && t < sb) w[(size_t)(k + 2 * hh) * sb + t] = val;   // final for the forward sweep
    }
    __syncthreads();
    // level 2h: w_k = w'_k - Q^T w'_{k-2h} - P^T w'_{k+2h}
    const double s = group_sum(mcol_dot(q2, w1[0], g) + mcol_dot(p2, w1[2], g), red[0], t, g);
    if (g == 0 && t < sb) w[(size_t)k * sb + t] = w1[1][t] - s;
}

// levels 2h then h of the backward sweep; own_lo = the right-hand side the level-h nodes read (b at h = 1, else w)
__global__ __launch_bounds__(kCrSolveThreads) void cr_bwd2_kernel(const CrView v, const uint32_t h, const double* __restrict__ own_lo,
                                                                  const uint32_t n_own_lo, const double* __restrict__ w,
                                                                  double* __restrict__ xs, double* __restrict__ x, const uint32_t n_x,
                                                                  const double* __restrict__ done_flag)
{
    __shared__ double red[kCrSolveGroups][64];
    __shared__ double yv[5][64];    // 0: w_i  1: x_{i-2h}  2: x_{i+2h}  3: own_j  4: x of j's other neighbour
    __shared__ double xi[64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const size_t sb2 = (size_t)sb * sb;
    const int64_t m = v.m, hh = h, j = hh * (2 * (int64_t)blockIdx.x + 1);
    if (j >= m) return;
    const int64_t lo = j - hh, hi = j + hh;
    const bool i_is_lo = (lo % (4 * hh)) == 2 * hh;
    const int64_t i = i_is_lo ? lo : hi, other = i_is_lo ? hi : lo;
    const bool i_ok = i < m, other_ok = other < m;
    const bool i_c = i_ok && i + 2 * hh < m;   // i - 2h >= 0 always
    const uint32_t n_w = v.m * (uint32_t)sb;
    MCol ei, pti, qti, ej, ptj, qtj;
    mcol_load(ei, v.E + (i_ok ? i : 0) * sb2, sb, t, g, i_ok);
    mcol_load(pti, v.PT + (i_ok ? i : 0) * sb2, sb, t, g, i_ok);
    mcol_load(qti, v.QT + (i_ok ? i : 0) * sb2, sb, t, g, i_c);
    mcol_load(ej, v.E + j * sb2, sb, t, g, true);
    mcol_load(ptj, v.PT + j * sb2, sb, t, g, true);
    mcol_load(qtj, v.QT + j * sb2, sb, t, g, hi < m);
    if (g < 5) {
        int64_t node; bool ok; const double* p = w; uint32_t lim = n_w;
        if (g == 0) { node = i; ok = i_ok; }                                    // right-hand side of i: forward value
        else if (g == 1) { node = i - 2 * hh; ok = i_ok; p = xs; }
        else if (g == 2) { node = i + 2 * hh; ok = i_c; p = xs; }
        else if (g == 3) { node = j; ok = true; p = own_lo; lim = n_own_lo; }
        else { node = other; ok = other_ok; p = xs; }
        const size_t idx = (size_t)(ok ? node : 0) * sb + t;
        yv[g][t] = (ok && t < sb && idx < lim) ? p[idx] : 0.0;
    }
    __syncthreads();
    // level 2h: x_i = E_i w_i - P_i x_{i-2h} - Q_i x_{i+2h}
    {
        const double s = group_sum(mcol_dot(ei, yv[0], g) - mcol_dot(pti, yv[1], g) - mcol_dot(qti, yv[2], g), red, t, g);
        if (g == 0) {
            xi[t] = (i_ok && t < sb) ? s : 0.0;
            if (i_ok && !i_is_lo && t < sb) {   // i = j + h: this workgroup owns x_i
                const size_t ii = (size_t)i * sb + t;
                xs[ii] = s;
                if (ii < n_x) x[ii] = s;
            }
        }
        __syncthreads();
    }
    // level h: x_j = E_j own_j - P_j x_{j-h} - Q_j x_{j+h}
    const double* xlo = i_is_lo ? xi : yv[4];
    const double* xhi = i_is_lo ? yv[4] : xi;
    const double s = group_sum(mcol_dot(ej, yv[3], g) - mcol_dot(ptj, xlo, g) - mcol_dot(qtj, xhi, g), red, t, g);
    if (g == 0 && t < sb) {
        const size_t jj = (size_t)j * sb + t;
        xs[jj] = s;
        if (jj < n_x) x[jj] = s;
    }
}

// The top of the tree in one launch.  At the largest stride h_top < m the forward sweep has one receiving node (0), then
// node 0 is solved, then the backward sweep has one node (h_top): three dependent matrix-vector products of single
// workgroups.  All four operators are requested up front, the vectors travel through LDS.  Arithmetic and its order
// are those of cr_fwd_kernel / cr_bwd_kernel.  m == 1: just x_0 = E_0 src_0.
__global__ __launch_bounds__(kCrSolveThreads) void cr_top_kernel(const CrView v, const uint32_t h_top, const double* __restrict__ src,
                                                                 const uint32_t n_src, double* __restrict__ xs, double* __restrict__ x,
                                                                 const uint32_t n_x, const double* __restrict__ done_flag)
{
    __shared__ double red[kCrSolveGroups][64];
    __shared__ double y0[64], yc[64], w0[64], x0[64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const size_t sb2 = (size_t)sb * sb;
    const bool has_c = h_top < v.m;    // false only for m == 1
    const uint32_t c = has_c ? h_top : 0;
    double mP[kCrSolveIters], mE0[kCrSolveIters], mEc[kCrSolveIters], mPT[kCrSolveIters];
#pragma unroll
    for (int u = 0; u < kCrSolveIters; u++) {
        const int m = g + u * kCrSolveGroups;
        const bool in = m < sb && t < sb;
        const size_t off = (size_t)m * sb + t;
        mP[u] = (in && has_c) ? v.P[c * sb2 + off] : 0.0;
        mE0[u] = in ? v.E[off] : 0.0;
        mEc[u] = (in && has_c) ? v.E[c * sb2 + off] : 0.0;
        mPT[u] = (in && has_c) ? v.PT[c * sb2 + off] : 0.0;
    }
    if (g == 0) {
        const size_t i0 = t, ic = (size_t)c * sb + t;
        y0[t] = (t < sb && i0 < n_src) ? src[i0] : 0.0;
        yc[t] = (has_c && t < sb && ic < n_src) ? src[ic] : 0.0;
    }
    __syncthreads();
    // forward: w_0 = src_0 - P_c^T src_c
    {
        double d = 0.0;
#pragma unroll
        for (int u = 0; u < kCrSolveIters; u++) { const int m = g + u * kCrSolveGroups; d += mP[u] * (m < sb ? yc[m] : 0.0); }
        const double s = group_sum(0.0 + d, red, t, g);
        if (g == 0) w0[t] = has_c ? y0[t] - s : y0[t];
        __syncthreads();
    }
    // last node: x_0 = E_0 w_0
    {
        double d = 0.0;
#pragma unroll
        for (int u = 0; u < kCrSolveIters; u++) { const int m = g + u * kCrSolveGroups; d += mE0[u] * (m < sb ? w0[m] : 0.0); }
        const double s = group_sum(d - 0.0 - 0.0, red, t, g);
        if (g == 0) {
            x0[t] = s;
            if (t < sb) {
                xs[t] = s;
                if ((uint32_t)t < n_x) x[t] = s;
            }
        }
        __syncthreads();
    }
    // backward: x_c = E_c src_c - P_c x_0
    if (has_c) {
        double de = 0.0, da = 0.0;
#pragma unroll
        for (int u = 0; u < kCrSolveIters; u++) {
            const int m = g + u * kCrSolveGroups;
            de += mEc[u] * (m < sb ? yc[m] : 0.0);
            da += mPT[u] * (m < sb ? x0[m] : 0.0);
        }
        const double s = group_sum(de - da - 0.0, red, t, g);
        if (g == 0 && t < sb) {
            const size_t ic = (size_t)c * sb + t;
            xs[ic] = s;
            if (ic < n_x) x[ic] = s;
        }
    }
}

// The top of the tree plus the level below it in one launch, for trees whose number of levels below the top is odd (the
// pairs above leave one): nodes 0, q = h_top / 2, 2q = h_top, 3q.  Forward level q (nodes 0 and 2q), forward level 2q
// (node 0), the root, backward level 2q (node 2q), backward level q (nodes q and 3q) - five dependent steps of one
// workgroup, all twelve operators requested up front.  Arithmetic and order as in cr_fwd / cr_top / cr_bwd.
__global__ __launch_bounds__(kCrSolveThreads) void cr_top2_kernel(const CrView v, const uint32_t q, const double* __restrict__ src,
                                                                  const uint32_t n_src, double* __restrict__ xs, double* __restrict__ x,
                                                                  const uint32_t n_x, const double* __restrict__ done_flag)
{
    __shared__ double red[2][kCrSolveGroups][64];
    __shared__ double y[4][64];      // src of 0, q, 2q, 3q
    __shared__ double w0a[64], w2a[64], w0b[64], x0[64], x2[64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t n1 = q, n2 = 2 * q, n3 = 3 * q;
    const bool has3 = n3 < v.m;      // q and 2q exist (2q = h_top < m)
    MCol pq, q1, p3, p2, e0, e2, pt2, e1, pt1, qt1, e3, pt3;
    mcol_load(pq, v.P + n1 * sb2, sb, t, g, true);           // forward q, node 0:  c = q
    mcol_load(q1, v.Q + n1 * sb2, sb, t, g, true);           // forward q, node 2q: a = q
    mcol_load(p3, v.P + (has3 ? n3 : 0) * sb2, sb, t, g, has3);   //                c = 3q
    mcol_load(p2, v.P + n2 * sb2, sb, t, g, true);           // forward 2q, node 0: c = 2q
    mcol_load(e0, v.E, sb, t, g, true);
    mcol_load(e2, v.E + n2 * sb2, sb, t, g, true);
    mcol_load(pt2, v.PT + n2 * sb2, sb, t, g, true);
    mcol_load(e1, v.E + n1 * sb2, sb, t, g, true);
    mcol_load(pt1, v.PT + n1 * sb2, sb, t, g, true);
    mcol_load(qt1, v.QT + n1 * sb2, sb, t, g, true);
    mcol_load(e3, v.E + (has3 ? n3 : 0) * sb2, sb, t, g, has3);
    mcol_load(pt3, v.PT + (has3 ? n3 : 0) * sb2, sb, t, g, has3);
    if (g < 4) {
        const uint32_t node = g * q;
        const size_t idx = (size_t)node * sb + t;
        y[g][t] = (node < v.m && t < sb && idx < n_src) ? src[idx] : 0.0;
    }
    __syncthreads();
    // forward level q: w_0 -= P_q^T w_q;  w_2q -= Q_q^T w_q + P_3q^T w_3q
    red[0][g][t] = 0.0 + mcol_dot(pq, y[1], g);
    red[1][g][t] = mcol_dot(q1, y[1], g) + mcol_dot(p3, y[3], g);
    __syncthreads();
    if (g < 2) {
        double s = red[g][0][t];
#pragma unroll
        for (int u = 1; u < kCrSolveGroups; u++) s += red[g][u][t];
        if (g == 0) w0a[t] = t < sb ? y[0][t] - s : 0.0;
        else w2a[t] = t < sb ? y[2][t] - s : 0.0;
    }
    __syncthreads();
    // forward level 2q: w_0 -= P_2q^T w_2q
    {
        const double s = group_sum(0.0 + mcol_dot(p2, w2a, g), red[0], t, g);
        if (g == 0) w0b[t] = t < sb ? w0a[t] - s : 0.0;
        __syncthreads();
    }
    // root
    {
        const double s = group_sum(mcol_dot(e0, w0b, g) - 0.0 - 0.0, red[0], t, g);
        if (g == 0) {
            x0[t] = t < sb ? s : 0.0;
            if (t < sb) {
                xs[t] = s;
                if ((uint32_t)t < n_x) x[t] = s;
            }
        }
        __syncthreads();
    }
    // backward level 2q: x_2q = E w_2q - P x_0
    {
        const double s = group_sum(mcol_dot(e2, w2a, g) - mcol_dot(pt2, x0, g) - 0.0, red[0], t, g);
        if (g == 0) {
            x2[t] = t < sb ? s : 0.0;
            if (t < sb) {
                const size_t i2 = (size_t)n2 * sb + t;
                xs[i2] = s;
                if (i2 < n_x) x[i2] = s;
            }
        }
        __syncthreads();
    }
    // backward level q: x_q = E w_q - P x_0 - Q x_2q;  x_3q = E w_3q - P x_2q
    red[0][g][t] = mcol_dot(e1, y[1], g) - mcol_dot(pt1, x0, g) - mcol_dot(qt1, x2, g);
    red[1][g][t] = mcol_dot(e3, y[3], g) - mcol_dot(pt3, x2, g) - 0.0;
    __syncthreads();
    if (g < 2 && t < sb && (g == 0 || has3)) {
        double s = red[g][0][t];
#pragma unroll
        for (int u = 1; u < kCrSolveGroups; u++) s += red[g][u][t];
        const size_t ii = (size_t)(g == 0 ? n1 : n3) * sb + t;
        xs[ii] = s;
        if (ii < n_x) x[ii] = s;
    }
}

constexpr int kCrMats = 8;

CrView make_view(uint32_t n_rows, int bw, double* ws)
{
    CrView v;
    v.bw = bw; v.sb = 6 * bw;
    v.m = (n_rows + (uint32_t)bw - 1) / (uint32_t)bw;
    v.n_real = n_rows * 6;
    const size_t per = (size_t)v.m * v.sb * v.sb;
    v.D = ws; v.F = ws + per; v.E = ws + 2 * per; v.P = ws + 3 * per; v.Q = ws + 4 * per; v.PT = ws + 5 * per; v.QT = ws + 6 * per;
    v.F2 = ws + 7 * per;
    return v;
}

uint32_t count_odd(uint32_t m, uint32_t h) { return m > h ? (m - h + 2 * h - 1) / (2 * h) : 0; }   // i = h(2j+1) < m
uint32_t count_even(uint32_t m, uint32_t h) { return (m + 2 * h - 1) / (2 * h); }                  // k = 2hj < m

}  // namespace

size_t cr_count(uint32_t n_rows, int bw)
{
    const size_t m = (n_rows + (size_t)bw - 1) / (size_t)bw, sb = 6 * (size_t)bw;
    return kCrMats * m * sb * sb + 3 * m * sb + 64;   // D F E P Q PT QT F2 + two padded right-hand-side buffers and the solution
}

size_t cr_map_count(uint32_t n_rows, int bw)
{
    const size_t m = (n_rows + (size_t)bw - 1) / (size_t)bw;
    return 2 * m * bw * bw;
}

// map[(i bw + la) bw + lb], first m super-blocks for D (la <= lb used), then m for F: the block of S that fills the 6x6
// block (camera la of super-block i, camera lb of super-block i resp. i - 1); -1: zero, -2: unit diagonal (padding)
void cr_build_map(uint32_t n_rows, int bw, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col, int32_t* map)
{
    const uint32_t m = (n_rows + (uint32_t)bw - 1) / (uint32_t)bw;
    const size_t per = (size_t)m * bw * bw;
    for (size_t e = 0; e < 2 * per; e++) map[e] = -1;
    for (uint32_t cam = n_rows; cam < m * (uint32_t)bw; cam++) {
        const uint32_t i = cam / (uint32_t)bw, l = cam - i * (uint32_t)bw;
        map[((size_t)i * bw + l) * bw + l] = -2;
    }
    for (uint32_t b = 0; b < n_blocks; b++) {
        const uint32_t ca = blk_row[b], cb = blk_col[b];   // ca <= cb
        const uint32_t ia = ca / (uint32_t)bw, ib = cb / (uint32_t)bw;
        const uint32_t la = ca - ia * (uint32_t)bw, lb = cb - ib * (uint32_t)bw;
        if (ia == ib) map[((size_t)ia * bw + la) * bw + lb] = (int32_t)b;
        else if (ib == ia + 1) map[per + ((size_t)ib * bw + lb) * bw + la] = (int32_t)b;   // F_ib: row camera cb, column camera ca
        // else: a block outside the block-tridiagonal part (a loop closure: the caller factors the band without it and keeps the
        // block in the matrix-vector product of its PCG)
    }
}

// levels the forward sweep of a solve runs two at a time (cr_fwd2): 2 * pairs levels from h = 1; the rest belongs to the top kernel
static int cr_fwd_pairs(uint32_t m)
{
    uint32_t h_top = 1;
    while (2 * h_top < m) h_top *= 2;
    int pairs = 0;
    for (uint32_t h = 1; 2 * h < h_top; h *= 4) pairs++;
    return pairs;
}

// comp[6 f + a] = sum over the blocks of S that the map leaves out (cr_build_map) of the absolute values in scalar row 6 f + a;
// lists: per camera the left-out blocks it belongs to (side 0: as their row camera, 1: as their column camera), host-built
__global__ __launch_bounds__(256) void cr_comp_kernel(const double* __restrict__ blocks, const uint32_t n6, const uint32_t* __restrict__ ptr,
                                                      const uint32_t* __restrict__ ent, double* __restrict__ comp)
{
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n6) return;
    const uint32_t f = u / 6, a = u % 6;
    double sum = 0.0;
    for (uint32_t e = ptr[f]; e < ptr[f + 1]; e++) {
        const uint32_t id = ent[e] >> 1, side = ent[e] & 1;
        const double* B = blocks + 36 * (size_t)id;
#pragma unroll
        for (int c = 0; c < 6; c++) sum += fabs(side ? B[c * 6 + a] : B[a * 6 + c]);
    }
    comp[u] = sum;
}

void launch_cr_comp(hipStream_t s, const double* blocks, uint32_t n_rows, const uint32_t* ptr, const uint32_t* ent, double* comp)
{
    if (!n_rows) return;
    hipLaunchKernelGGL(cr_comp_kernel, dim3((n_rows * 6 + 255) / 256), dim3(256), 0, s, blocks, n_rows * 6, ptr, ent, comp);
}

// host: the blocks cr_build_map leaves out, per camera (entry = 2 * block + side)
void cr_build_comp_lists(uint32_t n_rows, int bw, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col, std::vector<uint32_t>& ptr,
                         std::vector<uint32_t>& ent)
{
    std::vector<std::vector<uint32_t>> lists(n_rows);
    for (uint32_t b = 0; b < n_blocks; b++) {
        const uint32_t ia = blk_row[b] / (uint32_t)bw, ib = blk_col[b] / (uint32_t)bw;
        if (ib > ia + 1) { lists[blk_row[b]].push_back(2 * b); lists[blk_col[b]].push_back(2 * b + 1); }
    }
    ptr.assign(n_rows + 1, 0);
    ent.clear();
    for (uint32_t f = 0; f < n_rows; f++) { ptr[f + 1] = ptr[f] + (uint32_t)lists[f].size(); ent.insert(ent.end(), lists[f].begin(), lists[f].end()); }
}

void launch_cr_factor(hipStream_t s, const BsrView& A, const int32_t* map, int bw, double* ws, double* scal, const CamDamp* damp,
                      const double* fwd_b, const double* comp, double comp_scale)
{
    if (!A.n_rows) return;
    const CrView v = make_view(A.n_rows, bw, ws);
    CamDamp none{};
    hipLaunchKernelGGL(cr_gather_kernel, dim3(v.m, 2), dim3(256), 0, s, const_cast<double*>(A.blocks), map, v, damp ? *damp : none,
                       comp_scale > 0.0 ? comp : nullptr, comp_scale);
    const size_t lds_r = sizeof(double) * 4 * kImgRows * kLd;
    if (lds_r > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cr_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r);
    const uint32_t nt = ((uint32_t)v.sb + 15) / 16;
    static const char* const stg = std::getenv("SOSLAM_CR_STAGGER");      // test hook, see cr_reduce_kernel; at most 1 ms (read once)
    static const uint32_t stagger = stg ? (uint32_t)std::min<long>(std::max<long>(std::atol(stg), 0), 100000) : 0u;
    const double* fin = v.F;     // the gather's couplings; each level reads one coupling array and writes the other
    double* fout = v.F2;
    // the forward sweep of the solve that follows (fwd_b: its right-hand side), for the levels launch_cr_solve would run as pairs
    const int fwd_levels = fwd_b ? 2 * cr_fwd_pairs(v.m) : 0;
    double* W = ws + kCrMats * (size_t)v.m * v.sb * v.sb;
    const uint32_t n = A.n_rows * 6, n_pad = v.m * (uint32_t)v.sb;
    int lev = 0;
    for (uint32_t h = 1; h < v.m; h *= 2, lev++) {
        launch_cr_invert(s, count_odd(v.m, h), v, h, 0, scal);
        CrFwd fwd{nullptr, 0u, nullptr};
        if (lev < fwd_levels) fwd = CrFwd{lev == 0 ? fwd_b : W, lev == 0 ? n : n_pad, W};
        hipLaunchKernelGGL(cr_reduce_kernel, dim3(count_even(v.m, h), nt), dim3(kCrReduceThreads), lds_r, s, v, h, fin, fout, stagger, fwd);
        double* t = const_cast<double*>(fin); fin = fout; fout = t;
    }
    launch_cr_invert(s, 1, v, 0u, 1, scal);
}

void launch_cr_solve(hipStream_t s, uint32_t n_rows, int bw, double* ws, const double* b, double* x, const double* done_flag,
                     bool forward_done)
{
    if (!n_rows) return;
    const CrView v = make_view(n_rows, bw, ws);
    const uint32_t n = n_rows * 6, n_pad = v.m * (uint32_t)v.sb;
    // three padded vectors behind the matrices: two right-hand-side buffers (every forward launch reads one array and
    // writes the other - with fused levels a node's old value is still being read elsewhere when its new one is stored)
    // and the solution
    double* buf[2] = {ws + kCrMats * (size_t)v.m * v.sb * v.sb, nullptr};
    buf[1] = buf[0] + n_pad;
    double* xs = buf[1] + n_pad;
    uint32_t h_top = 1;
    while (2 * h_top < v.m) h_top *= 2;           // largest power of two below m (1 for m <= 2)
    // The right-hand side a node takes into the backward sweep is the value it had when its level was reached: own[l]
    // for the nodes eliminated at stride 2^l.  No copy in, no copy out: the first level reads the caller's b, every
    // backward kernel writes x.  Levels below the top go in pairs (cr_fwd2 / cr_bwd2), a left-over level alone.
    struct Src { const double* p; uint32_t n; };
    Src own[34];
    Src cur{b, n};
    int nb = 0, lev = 0;
    uint32_t h = 1;
    if (forward_done) {
        // the factorisation carried these levels (launch_cr_factor, fwd_b == b): every node's value lies in buf[0], in place,
        // except the nodes of the first level, which were only read
        const int pairs = cr_fwd_pairs(v.m);
        for (int l = 0; l < 2 * pairs; l++) own[l] = l == 0 ? Src{b, n} : Src{buf[0], n_pad};
        if (pairs > 0) cur = Src{buf[0], n_pad};
        for (int q = 0; q < pairs; q++) h *= 4;
        lev = 2 * pairs;
    } else {
        for (; 2 * h < h_top; h *= 4, lev += 2) {
            double* out = buf[nb]; nb ^= 1;
            own[lev] = cur;
            hipLaunchKernelGGL(cr_fwd2_kernel, dim3((v.m + 4 * h - 1) / (4 * h)), dim3(kCrSolveThreads), 0, s, v, h, cur.p, cur.n, out, done_flag);
            cur = Src{out, n_pad};
            own[lev + 1] = cur;
        }
    }
    // an odd number of levels below the top leaves one: it joins the top (cr_top2)
    if (h < h_top)
        hipLaunchKernelGGL(cr_top2_kernel, dim3(1), dim3(kCrSolveThreads), 0, s, v, h, cur.p, cur.n, xs, x, n, done_flag);
    else
        hipLaunchKernelGGL(cr_top_kernel, dim3(1), dim3(kCrSolveThreads), 0, s, v, h_top, cur.p, cur.n, xs, x, n, done_flag);
    while (h > 1) {
        h /= 4; lev -= 2;   // the pair (2h, h) = levels lev + 1, lev
        hipLaunchKernelGGL(cr_bwd2_kernel, dim3(count_odd(v.m, h)), dim3(kCrSolveThreads), 0, s, v, h, own[lev].p, own[lev].n,
                           own[lev + 1].p, xs, x, n, done_flag);
    }
}

// A (60 nb)^2, in place; ebuf: 2 x 3600 f64 of scratch; status[0] = 1 when a pivot tile was not positive definite
void launch_dense_spd_inverse60(hipStream_t s, double* A, int nb, double* ebuf, double* status)
{
    if (nb < 1) return;
    const size_t lds2 = sizeof(double) * 2 * kImgRows * kLd;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bgj_pivot_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bgj_row_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bgj_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        attr_set[dev] = true;
    }
    double* E[2] = {ebuf, ebuf + kTile * kTile};
    for (int K = 0; K <= nb; K++) {
        // inversion of pivot K beside the column step of pivot K - 1 (nb - 1 tiles and the pivot tile itself)
        const int Kp = K - 1;
        hipLaunchKernelGGL(bgj_pivot_kernel, dim3(Kp >= 0 ? 1 + nb : 1), dim3(256), lds2, s, A, nb, K, Kp, E[K & 1], E[(K & 1) ^ 1], status);
        if (K == nb) break;
        if (nb > 1) {
            hipLaunchKernelGGL(bgj_row_kernel, dim3(nb - 1), dim3(256), lds2, s, A, nb, K, E[K & 1]);
            hipLaunchKernelGGL(bgj_update_kernel, dim3(nb - 1, nb - 1), dim3(256), lds2, s, A, nb, K);
        }
    }
}

bool dense2_fits(uint32_t n_rows) { return n_rows >= 1 && n_rows <= 2 * (uint32_t)kDense2Cams; }

void launch_dense2_solve(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col, const double* b,
                         double* x, double* scal)
{
    if (!A.n_rows) return;
    const size_t lds = sizeof(double) * 4 * 64 * kLd;
    static bool attr_set[64] = {};     // per device: the attribute belongs to the function on that device
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense2_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(dense2_solve_kernel, dim3(1), dim3(kDense2Threads), lds, s, A, n_blocks, blk_row, blk_col, b, x, scal);
}

}  // namespace soslam
