// crsolve.hip - block cyclic reduction for the block-banded reduced camera matrix (gfx950).
//
// A band of half-width bw blocks is block TRIDIAGONAL in super-blocks of bw cameras (sb = 6 bw scalars).  The
// sequential band Cholesky needs n dependent block steps on one CU (latency-bound: ~2 us per step, 1.2 ms at 499
// cameras).  Cyclic reduction instead eliminates every other super-block of the current level IN PARALLEL, one
// workgroup per super-block, and halves the system per level: ceil(log2 m) levels of two kernels each
// (m = 56 super-blocks at BASELINE.json configs[2]).  Per eliminated node i with neighbours i-h, i+h:
//     D_i = L_i L_i^T,  Linv_i = L_i^-1,  U_i = Linv_i F_i,  V_i = Linv_i G_i^T      (F_i = A(i,i-h), G_i = A(i+h,i) = F_{i+h})
// per remaining node k:   D_k -= V_{k-h}^T V_{k-h} + U_{k+h}^T U_{k+h},   F_k <- -V_{k-h}^T U_{k-h}   (fill: A(k,k-2h))
// Storing the INVERSE factor turns every later triangular solve into a matrix-vector product, so a solve is
// 2 ceil(log2 m) + 1 small launches without any dependent chain.  All sums run in a fixed order: bitwise
// reproducible.  Positive definiteness is inherited by every Schur complement, so no pivoting is needed.
#include "linsolve.h"

#include "ba_kernels.h"

namespace soslam {

namespace {

constexpr int kCrThreads = 512;

__device__ __forceinline__ double rsqrt_nr(double s)
{
    double y = __builtin_amdgcn_rsq(s);
    y = y * (1.5 - 0.5 * s * y * y);
    y = y * (1.5 - 0.5 * s * y * y);
    return y;
}

// lower Cholesky factor and its inverse of the 6x6 block at A (row stride lda), in registers
__device__ __forceinline__ bool chol6_lds(const double* __restrict__ A, int lda, double (&L)[6][6], double (&Li)[6][6])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
#pragma unroll
        for (int i = j; i < 6; i++) {
            double s = A[i * lda + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                ok = ok && (s > 0.0);
                const double y = rsqrt_nr(s);
                Li[j][j] = y;
                L[j][j] = s * y;
            } else {
                L[i][j] = s * Li[j][j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = 0.0;
#pragma unroll
            for (int k = j; k < i; k++) s -= L[i][k] * Li[k][j];
            Li[i][j] = s * Li[i][i];
        }
    return ok;
}

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int kLd = 64;   // leading dimension of the zero-padded LDS images fed to the matrix cores (sb <= 60)

// One 16x16 tile of C = P^T Q on the f64 matrix cores: C[i0+i][j0+j] = sum_m P[m][i0+i] Q[m][j0+j], m < kp (multiple
// of 4), P and Q zero-padded LDS images with leading dimension kLd.  v_mfma_f64_16x16x4_f64 register layout
// (scripts/mfma_f64_layout.hip): lane l feeds A[i = l%16][k = l/16] and B[k = l/16][j = l%16] and receives
// D[i = 4 r + l/16][j = l%16] in accumulator register r.  Both operand reads are 16 consecutive doubles per k.
__device__ __forceinline__ double4_t mfma_ptq_tile(const double* __restrict__ P, const double* __restrict__ Q, int kp, int i0,
                                                   int j0, int lane)
{
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = P + (lane / 16) * kLd + i0 + lane % 16;
    const double* pb = Q + (lane / 16) * kLd + j0 + lane % 16;
    for (int k0 = 0; k0 < kp; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0 * kLd], pb[k0 * kLd], acc, 0, 0, 0);
    return acc;
}

struct CrView {
    uint32_t m;       // super-blocks
    int bw, sb;       // cameras per super-block, scalars per super-block
    double *D, *F, *Linv, *U, *V;   // each [m][sb*sb], row-major
};

// scatter the upper block-sparse S into D (diagonal super-blocks) and F (F_i = A(i, i-1)); pad with identity
__global__ __launch_bounds__(64) void cr_assemble_kernel(const double* __restrict__ blocks, const uint32_t* __restrict__ blk_row,
                                                         const uint32_t* __restrict__ blk_col, uint32_t n_blocks,
                                                         uint32_t n_rows, const CrView v)
{
    const int t = threadIdx.x;
    const size_t sb2 = (size_t)v.sb * v.sb;
    if (blockIdx.x < n_blocks) {
        if (t >= 36) return;
        const uint32_t ca = blk_row[blockIdx.x], cb = blk_col[blockIdx.x];   // ca <= cb
        const uint32_t ia = ca / (uint32_t)v.bw, ib = cb / (uint32_t)v.bw;
        const int la = (int)(ca - ia * v.bw), lb = (int)(cb - ib * v.bw);
        const int r = t / 6, c = t % 6;
        const double val = blocks[36 * (size_t)blockIdx.x + t];
        if (ia == ib) {
            v.D[ia * sb2 + (size_t)(la * 6 + r) * v.sb + lb * 6 + c] = val;
            v.D[ia * sb2 + (size_t)(lb * 6 + c) * v.sb + la * 6 + r] = val;
        } else {
            v.F[ib * sb2 + (size_t)(lb * 6 + c) * v.sb + la * 6 + r] = val;
        }
    } else {
        // padding cameras of the last super-block: unit diagonal
        const uint32_t cam = n_rows + (blockIdx.x - n_blocks);
        if (t < 6 && cam < v.m * (uint32_t)v.bw) {
            const uint32_t i = cam / (uint32_t)v.bw;
            const int l = (int)(cam - i * v.bw) * 6 + t;
            v.D[i * sb2 + (size_t)l * v.sb + l] = 1.0;
        }
    }
}

// Eliminated node i = h (2 j + 1)  (or node 0 when final != 0): Cholesky of D_i in LDS, inverse factor, U_i, V_i.
__global__ __launch_bounds__(kCrThreads) void cr_factor_kernel(const CrView v, const uint32_t h, const int final_node,
                                                               double* __restrict__ scal)
{
    extern __shared__ double lds[];
    const int sb = v.sb, bw = v.bw, tid = threadIdx.x;
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t i = final_node ? 0u : h * (2 * blockIdx.x + 1);
    if (i >= v.m) return;
    const bool has_left = !final_node && i >= h, has_right = !final_node && i + h < v.m;
    const int kp = (sb + 3) & ~3;  // summation length padded to the matrix cores' k = 4
    double* A = lds;               // sb*sb : D_i, then L_i (lower)
    double* X = A + sb2;           // sb*sb : Linv_i
    double* XT = X + sb2;          // kp*kLd : Linv_i^T, zero-padded
    double* T = XT + kp * kLd;     // kp*kLd : staged F_i / G_i^T, zero-padded
    double* LiD = T + kp * kLd;    // bw*36 : inverses of the diagonal 6x6 factor blocks
    double* Tb = LiD + bw * 36;    // bw*36 : scratch of the blocked inversion

    for (int e = tid; e < sb * sb; e += kCrThreads) { A[e] = v.D[i * sb2 + e]; X[e] = 0.0; }
    __syncthreads();

    // ---- blocked right-looking Cholesky, 6x6 blocks -------------------------------------------------------
    for (int kb = 0; kb < bw; kb++) {
        double L[6][6], Li[6][6];
        const bool ok = chol6_lds(A + (size_t)(kb * 6) * sb + kb * 6, sb, L, Li);
        if (!ok && tid == 0) scal[SC_LIN_STATUS] = 1.0;
        const int nb = bw - 1 - kb;
        __syncthreads();   // every lane has read A_kk
        if (tid < nb * 6) {
            double* a = A + (size_t)((kb + 1) * 6 + tid) * sb + kb * 6;   // row (kb+1)*6 + tid of the panel
            const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
            a[0] = a0 * Li[0][0];
            a[1] = a0 * Li[1][0] + a1 * Li[1][1];
            a[2] = a0 * Li[2][0] + a1 * Li[2][1] + a2 * Li[2][2];
            a[3] = a0 * Li[3][0] + a1 * Li[3][1] + a2 * Li[3][2] + a3 * Li[3][3];
            a[4] = a0 * Li[4][0] + a1 * Li[4][1] + a2 * Li[4][2] + a3 * Li[4][3] + a4 * Li[4][4];
            a[5] = a0 * Li[5][0] + a1 * Li[5][1] + a2 * Li[5][2] + a3 * Li[5][3] + a4 * Li[5][4] + a5 * Li[5][5];
        } else if (tid >= 128 && tid < 128 + 36) {
            const int e = tid - 128, r = e / 6, c = e % 6;
            double lv = 0.0, iv = 0.0;
#pragma unroll
            for (int rr = 0; rr < 6; rr++)
#pragma unroll
                for (int cc = 0; cc < 6; cc++)
                    if (rr == r && cc == c && cc <= rr) { lv = L[rr][cc]; iv = Li[rr][cc]; }
            A[(size_t)(kb * 6 + r) * sb + kb * 6 + c] = lv;
            LiD[kb * 36 + e] = iv;
        }
        __syncthreads();
        // trailing update: A_ij -= L_ik L_jk^T for block rows kb < jb <= ib; one lane per (pair, row)
        const int n_items = nb * (nb + 1) / 2 * 6;
        for (int item = tid; item < n_items; item += kCrThreads) {
            const int pair = item / 6, r = item % 6;
            int ii = (int)((sqrtf(8.0f * (float)pair + 1.0f) - 1.0f) * 0.5f);
            while ((ii + 1) * (ii + 2) / 2 <= pair) ii++;
            while (ii * (ii + 1) / 2 > pair) ii--;
            const int jj = pair - ii * (ii + 1) / 2;
            const int ib = kb + 1 + ii, jb = kb + 1 + jj;
            const double* lik = A + (size_t)(ib * 6 + r) * sb + kb * 6;
            double* tgt = A + (size_t)(ib * 6 + r) * sb + jb * 6;
            const double l0 = lik[0], l1 = lik[1], l2 = lik[2], l3 = lik[3], l4 = lik[4], l5 = lik[5];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double* lj = A + (size_t)(jb * 6 + c) * sb + kb * 6;
                tgt[c] -= l0 * lj[0] + l1 * lj[1] + l2 * lj[2] + l3 * lj[3] + l4 * lj[4] + l5 * lj[5];
            }
        }
        __syncthreads();
    }

    // ---- blocked inversion of the lower factor: X[rb][cb] = -Li_rb * sum_{kb=cb}^{rb-1} L[rb][kb] X[kb][cb] ---------
    for (int rb = 0; rb < bw; rb++) {
        if (tid < 36) X[(size_t)(rb * 6 + tid / 6) * sb + rb * 6 + tid % 6] = LiD[rb * 36 + tid];
        for (int e = tid; e < rb * 36; e += kCrThreads) {
            const int cb = e / 36, r = (e % 36) / 6, c = e % 6;
            double s = 0.0;
            for (int kb = cb; kb < rb; kb++) {
                const double* l = A + (size_t)(rb * 6 + r) * sb + kb * 6;
                const double* x = X + (size_t)(kb * 6) * sb + cb * 6 + c;
#pragma unroll
                for (int m = 0; m < 6; m++) s += l[m] * x[(size_t)m * sb];
            }
            Tb[e] = s;
        }
        __syncthreads();
        for (int e = tid; e < rb * 36; e += kCrThreads) {
            const int cb = e / 36, r = (e % 36) / 6, c = e % 6;
            const double* li = LiD + rb * 36 + r * 6;
            const double* t = Tb + cb * 36 + c;
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 6; m++) s += li[m] * t[m * 6];
            X[(size_t)(rb * 6 + r) * sb + cb * 6 + c] = -s;
        }
        __syncthreads();
    }
    for (int e = tid; e < sb * sb; e += kCrThreads) v.Linv[i * sb2 + e] = X[e];

    // ---- U_i = Linv F_i and V_i = Linv G_i^T on the matrix cores: out = (Linv^T)^T T ---------------------------
    for (int e = tid; e < kp * kLd; e += kCrThreads) {
        const int k = e / kLd, r = e % kLd;
        XT[e] = (k < sb && r < sb) ? X[(size_t)r * sb + k] : 0.0;
    }
    for (int side = 0; side < 2; side++) {
        const bool on = side == 0 ? has_left : has_right;
        if (!on) continue;   // uniform
        __syncthreads();
        const double* src = side == 0 ? v.F + i * sb2 : v.F + (i + h) * sb2;   // G_i = A(i+h, i) = F_{i+h}, used transposed
        for (int e = tid; e < kp * kLd; e += kCrThreads) {
            const int k = e / kLd, c = e % kLd;
            T[e] = (k < sb && c < sb) ? (side == 0 ? src[(size_t)k * sb + c] : src[(size_t)c * sb + k]) : 0.0;
        }
        __syncthreads();
        double* out = (side == 0 ? v.U : v.V) + i * sb2;
        const int wave = tid / 64, lane = tid % 64;
        const int nt = (sb + 15) / 16;
        for (int t = wave; t < nt * nt; t += kCrThreads / 64) {
            const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
            // Linv is lower triangular: rows i0..i0+15 only reach k <= i0 + 15
            const double4_t acc = mfma_ptq_tile(XT, T, min(kp, (i0 + 16 + 3) & ~3), i0, j0, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i0 + 4 * r + lane / 16, col = j0 + lane % 16;
                if (row < sb && col < sb) out[(size_t)row * sb + col] = acc[r];
            }
        }
    }
}

// Remaining node k = 2 h j: D_k -= V_a^T V_a + U_c^T U_c (a = k-h, c = k+h), F_k <- -V_a^T U_a
__global__ __launch_bounds__(kCrThreads) void cr_update_kernel(const CrView v, const uint32_t h)
{
    extern __shared__ double lds[];
    const int sb = v.sb, tid = threadIdx.x;
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t k = 2 * h * blockIdx.x;
    if (k >= v.m) return;
    const bool has_a = k >= h, has_c = k + h < v.m;
    if (!has_a && !has_c) return;
    const int kp = (sb + 3) & ~3;
    double* P = lds;                 // kp*kLd : V_a or U_c, zero-padded
    double* Q = P + kp * kLd;        // kp*kLd : U_a
    const int wave = tid / 64, lane = tid % 64;
    const int nt = (sb + 15) / 16;
    auto stage = [&](double* dst, const double* src) {
        for (int e = tid; e < kp * kLd; e += kCrThreads) {
            const int r = e / kLd, c = e % kLd;
            dst[e] = (r < sb && c < sb) ? src[(size_t)r * sb + c] : 0.0;
        }
    };
    if (has_a) {
        const uint32_t a = k - h;
        const bool fill = a >= h;   // node a had a left neighbour a-h = k-2h: new coupling A(k, k-2h)
        stage(P, v.V + a * sb2);
        if (fill) stage(Q, v.U + a * sb2);
        __syncthreads();
        for (int t = wave; t < nt * nt; t += kCrThreads / 64) {
            const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
            const double4_t s = mfma_ptq_tile(P, P, kp, i0, j0, lane);
            double4_t f = {0.0, 0.0, 0.0, 0.0};
            if (fill) f = mfma_ptq_tile(P, Q, kp, i0, j0, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i0 + 4 * r + lane / 16, col = j0 + lane % 16;
                if (row < sb && col < sb) {
                    v.D[k * sb2 + (size_t)row * sb + col] -= s[r];
                    if (fill) v.F[k * sb2 + (size_t)row * sb + col] = -f[r];
                }
            }
        }
        __syncthreads();
    }
    if (has_c) {
        stage(P, v.U + (k + h) * sb2);
        __syncthreads();
        for (int t = wave; t < nt * nt; t += kCrThreads / 64) {
            const int i0 = (t / nt) * 16, j0 = (t % nt) * 16;
            const double4_t s = mfma_ptq_tile(P, P, kp, i0, j0, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i0 + 4 * r + lane / 16, col = j0 + lane % 16;
                if (row < sb && col < sb) v.D[k * sb2 + (size_t)row * sb + col] -= s[r];
            }
        }
    }
}

// ---- solve phase: every step is a matrix-vector product with a stored sb x sb operator ------------------------
// w holds the right-hand side on entry (padded to m*sb), the solution on exit.

__device__ __forceinline__ bool cr_done(const double* __restrict__ done_flag) { return done_flag && *done_flag != 0.0; }

// The solve kernels use 256 lanes per node: lane (t = tid % 64, g = tid / 64) sums a quarter of the products of
// output element t, the quarters are added in a fixed order through LDS.  Four times the loads in flight of a
// one-lane-per-element loop - these kernels are pure latency.
__device__ __forceinline__ double quarter_sum(double part, double (*red)[64], int t, int g)
{
    red[g][t] = part;
    __syncthreads();
    return red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// forward, eliminated node i: y_i = Linv_i w_i   (in place)
__global__ __launch_bounds__(256) void cr_fwd_kernel(const CrView v, const uint32_t h, const int final_node, double* __restrict__ w,
                                                     const double* __restrict__ done_flag)
{
    __shared__ double wl[64];
    __shared__ double red[4][64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = threadIdx.x / 64;
    const uint32_t i = final_node ? 0u : h * (2 * blockIdx.x + 1);
    if (i >= v.m) return;
    if (threadIdx.x < sb) wl[threadIdx.x] = w[(size_t)i * sb + threadIdx.x];
    __syncthreads();
    double s = 0.0;
    if (t < sb) {
        const double* M = v.Linv + (size_t)i * sb * sb + (size_t)t * sb;   // row t, columns 0..t
        for (int k = g; k <= t; k += 4) s += M[k] * wl[k];
    }
    s = quarter_sum(s, red, t, g);
    if (g == 0 && t < sb) w[(size_t)i * sb + t] = s;
}

// forward, remaining node k: w_k -= V_a^T y_a + U_c^T y_c
__global__ __launch_bounds__(256) void cr_fwd_update_kernel(const CrView v, const uint32_t h, double* __restrict__ w,
                                                            const double* __restrict__ done_flag)
{
    __shared__ double red[4][64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = threadIdx.x / 64;
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t k = 2 * h * blockIdx.x;
    if (k >= v.m) return;
    double s = 0.0;
    if (t < sb) {
        if (k >= h) {
            const uint32_t a = k - h;
            const double* M = v.V + a * sb2 + t;
            const double* y = w + (size_t)a * sb;
            for (int m = g; m < sb; m += 4) s += M[(size_t)m * sb] * y[m];
        }
        if (k + h < v.m) {
            const uint32_t c = k + h;
            const double* M = v.U + c * sb2 + t;
            const double* y = w + (size_t)c * sb;
            for (int m = g; m < sb; m += 4) s += M[(size_t)m * sb] * y[m];
        }
    }
    s = quarter_sum(s, red, t, g);
    if (g == 0 && t < sb) w[(size_t)k * sb + t] -= s;
}

// backward, node i eliminated at stride h (or the last node): x_i = Linv_i^T (y_i - U_i x_{i-h} - V_i x_{i+h})
__global__ __launch_bounds__(256) void cr_bwd_kernel(const CrView v, const uint32_t h, const int final_node, double* __restrict__ w,
                                                     const double* __restrict__ done_flag)
{
    __shared__ double tl[64];
    __shared__ double red[4][64];
    if (cr_done(done_flag)) return;
    const int sb = v.sb, t = threadIdx.x % 64, g = threadIdx.x / 64;
    const size_t sb2 = (size_t)sb * sb;
    const uint32_t i = final_node ? 0u : h * (2 * blockIdx.x + 1);
    if (i >= v.m) return;
    double s = 0.0;
    if (t < sb) {
        if (!final_node && i >= h) {
            const double* M = v.U + i * sb2 + (size_t)t * sb;
            const double* x = w + (size_t)(i - h) * sb;
            for (int m = g; m < sb; m += 4) s += M[m] * x[m];
        }
        if (!final_node && i + h < v.m) {
            const double* M = v.V + i * sb2 + (size_t)t * sb;
            const double* x = w + (size_t)(i + h) * sb;
            for (int m = g; m < sb; m += 4) s += M[m] * x[m];
        }
    }
    s = quarter_sum(s, red, t, g);
    if (g == 0 && t < sb) tl[t] = w[(size_t)i * sb + t] - s;
    __syncthreads();
    double u = 0.0;
    if (t < sb) {
        const double* M = v.Linv + i * sb2 + t;   // column t of Linv, rows t..sb-1
        for (int k = t + g; k < sb; k += 4) u += M[(size_t)k * sb] * tl[k];
    }
    u = quarter_sum(u, red, t, g);
    if (g == 0 && t < sb) w[(size_t)i * sb + t] = u;
}

__global__ __launch_bounds__(256) void cr_load_rhs_kernel(uint32_t n, uint32_t n_pad, const double* __restrict__ b, double* __restrict__ w,
                                                          const double* __restrict__ done_flag)
{
    if (cr_done(done_flag)) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_pad) w[i] = i < n ? b[i] : 0.0;
}

__global__ __launch_bounds__(256) void cr_store_x_kernel(uint32_t n, const double* __restrict__ w, double* __restrict__ x,
                                                         const double* __restrict__ done_flag)
{
    if (cr_done(done_flag)) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = w[i];
}

CrView make_view(uint32_t n_rows, int bw, double* ws)
{
    CrView v;
    v.bw = bw; v.sb = 6 * bw;
    v.m = (n_rows + (uint32_t)bw - 1) / (uint32_t)bw;
    const size_t per = (size_t)v.m * v.sb * v.sb;
    v.D = ws; v.F = ws + per; v.Linv = ws + 2 * per; v.U = ws + 3 * per; v.V = ws + 4 * per;
    return v;
}

uint32_t count_odd(uint32_t m, uint32_t h) { return m > h ? (m - h + 2 * h - 1) / (2 * h) : 0; }   // i = h(2j+1) < m
uint32_t count_even(uint32_t m, uint32_t h) { return (m + 2 * h - 1) / (2 * h); }                  // k = 2hj < m

}  // namespace

size_t cr_count(uint32_t n_rows, int bw)
{
    const size_t m = (n_rows + (size_t)bw - 1) / (size_t)bw, sb = 6 * (size_t)bw;
    return 5 * m * sb * sb + m * sb + 64;   // D F Linv U V + padded work vector
}

void launch_cr_factor(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col, int bw,
                      double* ws, double* scal)
{
    if (!A.n_rows) return;
    const CrView v = make_view(A.n_rows, bw, ws);
    const size_t per = (size_t)v.m * v.sb * v.sb;
    (void)hipMemsetAsync(v.D, 0, sizeof(double) * 2 * per, s);   // D and F
    const uint32_t pad = v.m * (uint32_t)bw - A.n_rows;
    hipLaunchKernelGGL(cr_assemble_kernel, dim3(n_blocks + pad), dim3(64), 0, s, A.blocks, blk_row, blk_col, n_blocks, A.n_rows, v);
    const size_t kp = ((size_t)v.sb + 3) & ~(size_t)3;
    const size_t lds_f = sizeof(double) * (2 * (size_t)v.sb * v.sb + 2 * kp * kLd + 2 * (size_t)bw * 36);
    const size_t lds_u = sizeof(double) * 2 * kp * kLd;
    if (lds_f > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cr_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    if (lds_u > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cr_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_u);
    for (uint32_t h = 1; h < v.m; h *= 2) {
        hipLaunchKernelGGL(cr_factor_kernel, dim3(count_odd(v.m, h)), dim3(kCrThreads), lds_f, s, v, h, 0, scal);
        hipLaunchKernelGGL(cr_update_kernel, dim3(count_even(v.m, h)), dim3(kCrThreads), lds_u, s, v, h);
    }
    hipLaunchKernelGGL(cr_factor_kernel, dim3(1), dim3(kCrThreads), lds_f, s, v, 0u, 1, scal);
}

void launch_cr_solve(hipStream_t s, uint32_t n_rows, int bw, double* ws, const double* b, double* x, const double* done_flag)
{
    if (!n_rows) return;
    const CrView v = make_view(n_rows, bw, ws);
    double* w = ws + 5 * (size_t)v.m * v.sb * v.sb;
    const uint32_t n = n_rows * 6, n_pad = v.m * (uint32_t)v.sb;
    hipLaunchKernelGGL(cr_load_rhs_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, s, n, n_pad, b, w, done_flag);
    uint32_t h = 1;
    for (; h < v.m; h *= 2) {
        hipLaunchKernelGGL(cr_fwd_kernel, dim3(count_odd(v.m, h)), dim3(256), 0, s, v, h, 0, w, done_flag);
        hipLaunchKernelGGL(cr_fwd_update_kernel, dim3(count_even(v.m, h)), dim3(256), 0, s, v, h, w, done_flag);
    }
    hipLaunchKernelGGL(cr_fwd_kernel, dim3(1), dim3(256), 0, s, v, 0u, 1, w, done_flag);
    hipLaunchKernelGGL(cr_bwd_kernel, dim3(1), dim3(256), 0, s, v, 0u, 1, w, done_flag);
    for (h /= 2; h >= 1; h /= 2) {
        hipLaunchKernelGGL(cr_bwd_kernel, dim3(count_odd(v.m, h)), dim3(256), 0, s, v, h, 0, w, done_flag);
        if (h == 1) break;
    }
    hipLaunchKernelGGL(cr_store_x_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, w, x, done_flag);
}

}  // namespace soslam
