// linsolve.hip - reduced camera system solvers for gfx950: block-Jacobi PCG on the block-sparse S and a
// blocked right-looking dense Cholesky.  The systems are small (<= a few thousand unknowns; 2.7 MB of
// blocks at BASELINE.json configs[2]), so both are latency-bound, not HBM-bound; they run as a handful of
// workgroups and keep the host out of the loop (no per-iteration synchronisation).
#include "linsolve.h"

#include "ba_kernels.h"
#include "reduce.h"

namespace soslam {

namespace {

constexpr int kWave = 64;
constexpr int kNB = 32;  // dense Cholesky panel width

// sum over a 1024-lane workgroup, identical (fixed-order) result in every lane
__device__ __forceinline__ double block_sum(double v, double* red)
{
    v = wave_sum(v);
    __syncthreads();
    if (threadIdx.x % kWave == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) t += red[w];
    return t;
}

// inverse of a symmetric positive definite 6x6 through its Cholesky factor; false if not SPD
__device__ bool spd6_inverse(const double* __restrict__ A, double* __restrict__ out)
{
    double L[6][6], Li[6][6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
#pragma unroll
        for (int i = j; i < 6; i++) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) {
                ok = ok && (s > 0.0);
                L[j][j] = sqrt(s);
            } else {
                L[i][j] = s / L[j][j];
            }
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

// y_row = sum over the block row's entries of (block or block^T)[a, :] . v[col]
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

// One workgroup runs the whole PCG; vectors live in global memory (L1/L2 resident), reductions are
// fixed-order so the result is bitwise reproducible.
__global__ __launch_bounds__(1024) void pcg_kernel(const BsrView A, const double* __restrict__ b, double* __restrict__ x,
                                                   double* __restrict__ r, double* __restrict__ work, const double tol,
                                                   const int max_iter, double* __restrict__ scal)
{
    __shared__ double red[16];
    const uint32_t n = A.n_rows * 6;
    double* p = work;
    double* z = work + n;
    double* q = work + 2 * (size_t)n;
    double* Minv = work + 4 * (size_t)n;
    const int tid = threadIdx.x;

    bool ok = true;
    for (uint32_t f = tid; f < A.n_rows; f += 1024)
        ok = spd6_inverse(A.blocks + 36 * (size_t)A.diag_block[f], Minv + 36 * (size_t)f) && ok;
    if (!ok) scal[SC_LIN_STATUS] = 1.0;
    __syncthreads();

    double bb = 0.0, rz = 0.0;
    for (uint32_t i = tid; i < n; i += 1024) {
        const double bi = b[i];
        x[i] = 0.0;
        r[i] = bi;
        bb += bi * bi;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) {
        const double* M = Minv + 36 * (size_t)(i / 6) + 6 * (i % 6);
        const double* rv = r + 6 * (size_t)(i / 6);
        const double zi = M[0] * rv[0] + M[1] * rv[1] + M[2] * rv[2] + M[3] * rv[3] + M[4] * rv[4] + M[5] * rv[5];
        z[i] = zi;
        p[i] = zi;
        rz += r[i] * zi;
    }
    bb = block_sum(bb, red);
    rz = block_sum(rz, red);
    const double stop2 = tol * tol * bb;
    double rr = bb;
    int it = 0;
    while (it < max_iter && rr > stop2 && rr > 0.0) {
        double pq = 0.0;
        for (uint32_t i = tid; i < n; i += 1024) {
            const double qi = bsr_row_dot(A, i, p);
            q[i] = qi;
            pq += p[i] * qi;
        }
        pq = block_sum(pq, red);
        if (!(pq > 0.0)) { if (tid == 0) scal[SC_LIN_STATUS] = 2.0; break; }
        const double alpha = rz / pq;
        double rr_new = 0.0;
        for (uint32_t i = tid; i < n; i += 1024) {
            x[i] += alpha * p[i];
            const double ri = r[i] - alpha * q[i];
            r[i] = ri;
            rr_new += ri * ri;
        }
        rr = block_sum(rr_new, red);  // the barrier inside also publishes r before z reads it
        double rz_new = 0.0;
        for (uint32_t i = tid; i < n; i += 1024) {
            const double* M = Minv + 36 * (size_t)(i / 6) + 6 * (i % 6);
            const double* rv = r + 6 * (size_t)(i / 6);
            const double zi = M[0] * rv[0] + M[1] * rv[1] + M[2] * rv[2] + M[3] * rv[3] + M[4] * rv[4] + M[5] * rv[5];
            z[i] = zi;
            rz_new += r[i] * zi;
        }
        rz_new = block_sum(rz_new, red);
        const double beta = rz_new / rz;
        rz = rz_new;
        for (uint32_t i = tid; i < n; i += 1024) p[i] = z[i] + beta * p[i];
        __syncthreads();
        it++;
    }
    if (tid == 0) {
        scal[SC_LIN_ITERS] = (double)it;
        scal[SC_LIN_RESID] = bb > 0.0 ? sqrt(rr / bb) : 0.0;
    }
}

// ---- dense path --------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void bsr_to_dense_kernel(const double* __restrict__ blocks, const uint32_t* __restrict__ blk_row,
                                                          const uint32_t* __restrict__ blk_col, uint32_t n,
                                                          double* __restrict__ dense)
{
    const uint32_t blk = blockIdx.x;
    const int t = threadIdx.x;
    if (t >= 36) return;
    const uint32_t i = 6 * blk_row[blk] + t / 6, j = 6 * blk_col[blk] + t % 6;
    const double v = blocks[36 * (size_t)blk + t];
    dense[(size_t)i * n + j] = v;
    dense[(size_t)j * n + i] = v;
}

// factor the nb x nb diagonal block at k0 in LDS, write L back, and store inv(L) for the solves
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, uint32_t n, uint32_t k0,
                                                        double* __restrict__ dinv, double* __restrict__ scal)
{
    __shared__ double T[kNB][kNB + 1];
    __shared__ double Ti[kNB][kNB + 1];
    const int nb = min((uint32_t)kNB, n - k0);
    const int tid = threadIdx.x;
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int i = e / kNB, j = e % kNB;
        T[i][j] = (i < nb && j < nb && j <= i) ? A[(size_t)(k0 + i) * n + k0 + j] : (i == j ? 1.0 : 0.0);
        Ti[i][j] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < nb; j++) {
        if (tid == 0) {
            const double d = T[j][j];
            if (!(d > 0.0)) scal[SC_LIN_STATUS] = 1.0;
            T[j][j] = sqrt(d);
        }
        __syncthreads();
        const double djj = T[j][j];
        if (tid > j && tid < nb) T[tid][j] /= djj;
        __syncthreads();
        for (int e = tid; e < kNB * kNB; e += 256) {
            const int i = e / kNB, k = e % kNB;
            if (k > j && i >= k && i < nb) T[i][k] -= T[i][j] * T[k][j];
        }
        __syncthreads();
    }
    // inverse of the lower-triangular factor: lane j solves column j by forward substitution
    if (tid < kNB) {
        const int j = tid;
        for (int i = j; i < kNB; i++) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; k++) s -= T[i][k] * Ti[k][j];
            Ti[i][j] = s / T[i][i];
        }
    }
    __syncthreads();
    for (int e = tid; e < kNB * kNB; e += 256) {
        const int i = e / kNB, j = e % kNB;
        if (i < nb && j <= i) A[(size_t)(k0 + i) * n + k0 + j] = T[i][j];
        dinv[(size_t)(k0 / kNB) * kNB * kNB + e] = Ti[i][j];
    }
}

// rows below the diagonal block: L_ik = A_ik L_kk^-T, one lane per row, panel width is always kNB here
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, uint32_t n, uint32_t k0)
{
    __shared__ double Lkk[kNB][kNB + 1];
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) {
        const int i = e / kNB, j = e % kNB;
        Lkk[i][j] = (j <= i) ? A[(size_t)(k0 + i) * n + k0 + j] : 0.0;
    }
    __syncthreads();
    const uint32_t i = k0 + kNB + blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* row = A + (size_t)i * n + k0;
    double x[kNB];
#pragma unroll
    for (int j = 0; j < kNB; j++) x[j] = row[j];
#pragma unroll
    for (int j = 0; j < kNB; j++) {
        double s = x[j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= x[k] * Lkk[j][k];
        x[j] = s / Lkk[j][j];
    }
#pragma unroll
    for (int j = 0; j < kNB; j++) row[j] = x[j];
}

// trailing update of the lower triangle: A_ij -= L_i L_j^T over the panel, 32x32 tiles
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, uint32_t n, uint32_t k0)
{
    if (blockIdx.y > blockIdx.x) return;
    __shared__ double Li[kNB][kNB + 1];
    __shared__ double Lj[kNB][kNB + 1];
    const uint32_t base = k0 + kNB;
    const uint32_t i0 = base + blockIdx.x * kNB, j0 = base + blockIdx.y * kNB;
    for (int e = threadIdx.x; e < kNB * kNB; e += 256) {
        const int r = e / kNB, c = e % kNB;
        Li[r][c] = (i0 + r < n) ? A[(size_t)(i0 + r) * n + k0 + c] : 0.0;
        Lj[r][c] = (j0 + r < n) ? A[(size_t)(j0 + r) * n + k0 + c] : 0.0;
    }
    __syncthreads();
    const int c = threadIdx.x % kNB, r4 = threadIdx.x / kNB;  // 8 row groups x 4 rows
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int r = r4 * 4 + rr;
        const uint32_t i = i0 + r, j = j0 + c;
        if (i < n && j < n && j <= i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < kNB; k++) s += Li[r][k] * Lj[c][k];
            A[(size_t)i * n + j] -= s;
        }
    }
}

// L y = b then L^T x = y with the stored inverses of the diagonal blocks; one workgroup
__global__ __launch_bounds__(1024) void chol_solve_kernel(const double* __restrict__ A, uint32_t n,
                                                          const double* __restrict__ dinv, const double* __restrict__ b,
                                                          double* __restrict__ x)
{
    __shared__ double yb[kNB];
    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < n; i += 1024) x[i] = b[i];
    __syncthreads();
    const uint32_t nblk = (n + kNB - 1) / kNB;
    for (uint32_t kb = 0; kb < nblk; kb++) {
        const uint32_t k0 = kb * kNB;
        const int nb = min((uint32_t)kNB, n - k0);
        const double* Di = dinv + (size_t)kb * kNB * kNB;
        if (tid < nb) {
            double s = 0.0;
            for (int k = 0; k <= tid; k++) s += Di[tid * kNB + k] * x[k0 + k];
            yb[tid] = s;
        }
        __syncthreads();
        if (tid < nb) x[k0 + tid] = yb[tid];
        for (uint32_t i = k0 + nb + tid; i < n; i += 1024) {
            const double* row = A + (size_t)i * n + k0;
            double s = 0.0;
            for (int k = 0; k < nb; k++) s += row[k] * yb[k];
            x[i] -= s;
        }
        __syncthreads();
    }
    for (uint32_t kb = nblk; kb-- > 0;) {
        const uint32_t k0 = kb * kNB;
        const int nb = min((uint32_t)kNB, n - k0);
        const double* Di = dinv + (size_t)kb * kNB * kNB;
        if (tid < nb) {
            double s = 0.0;
            for (int k = tid; k < nb; k++) s += Di[k * kNB + tid] * x[k0 + k];
            yb[tid] = s;
        }
        __syncthreads();
        if (tid < nb) x[k0 + tid] = yb[tid];
        for (uint32_t k = tid; k < k0; k += 1024) {
            double s = 0.0;
            for (int i = 0; i < nb; i++) s += A[(size_t)(k0 + i) * n + k] * yb[i];
            x[k] -= s;
        }
        __syncthreads();
    }
}

}  // namespace

size_t pcg_work_count(uint32_t n_rows) { return (size_t)n_rows * (4 * 6 + 36); }

void launch_pcg(hipStream_t s, const BsrView& A, const double* b, double* x, double* resid, double* work,
                double tol, int max_iter, double* scal)
{
    if (!A.n_rows) return;
    hipLaunchKernelGGL(pcg_kernel, dim3(1), dim3(1024), 0, s, A, b, x, resid, work, tol, max_iter, scal);
}

// ---- small dense systems: everything in one workgroup's LDS -------------------------------------------------
// The sliding windows of the reference (slam.cpp:121-129: at most 2 x refine_interval = 20 frames, 19 of them free)
// give a reduced system of at most 114 unknowns.  One workgroup expands the block-sparse S into LDS, factors it by a
// right-looking blocked Cholesky (6x6 blocks: wave 0 factors the diagonal block, one lane per panel row, one lane
// per (block pair, row) of the trailing update) and runs both substitutions, 3 barriers per block step - no global
// round trip between the steps, which is all the blocked multi-kernel path above spends its time on at this size.
namespace {

constexpr int kSmallThreads = 1024;

__device__ __forceinline__ double small_rsqrt(double s)
{
    double y = __builtin_amdgcn_rsq(s);
    y = y * (1.5 - 0.5 * s * y * y);
    y = y * (1.5 - 0.5 * s * y * y);
    return y;
}

__global__ __launch_bounds__(kSmallThreads) void dense_small_solve_kernel(const BsrView A, const uint32_t n_blocks,
                                                                          const uint32_t* __restrict__ blk_row,
                                                                          const uint32_t* __restrict__ blk_col,
                                                                          const double* __restrict__ b, double* __restrict__ x,
                                                                          double* __restrict__ scal)
{
    extern __shared__ double lds[];
    const int nb = (int)A.n_rows, n = 6 * nb, ld = n | 1, tid = threadIdx.x;
    double* M = lds;                    // n x ld, lower triangle becomes L
    double* Li = M + (size_t)n * ld;    // nb x 36: inverses of the diagonal factor blocks
    double* v = Li + nb * 36;           // n: right-hand side -> y -> x

    for (int e = tid; e < n * ld; e += kSmallThreads) M[e] = 0.0;
    if (tid < n) v[tid] = b[tid];
    __syncthreads();
    for (uint32_t e = tid; e < n_blocks * 36; e += kSmallThreads) {
        const uint32_t blk = e / 36, t = e - blk * 36;
        const int r = (int)t / 6, c = (int)t % 6;
        const int i = 6 * (int)blk_row[blk] + r, j = 6 * (int)blk_col[blk] + c;   // blk_row <= blk_col
        const double val = A.blocks[e];
        M[j * ld + i] = val;            // lower triangle (and the diagonal blocks whole)
        if (blk_row[blk] == blk_col[blk]) M[i * ld + j] = val;
    }
    __syncthreads();

    for (int kb = 0; kb < nb; kb++) {
        const int k0 = 6 * kb;
        if (tid < 64) {
            // diagonal block: L_kk and its inverse, lane 0 of wave 0 (a 6x6 chain; the other lanes would only repeat it)
            if (tid == 0) {
                double L[6][6], I[6][6];
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 6; j++) {
#pragma unroll
                    for (int i = j; i < 6; i++) {
                        double sacc = M[(k0 + i) * ld + k0 + j];
#pragma unroll
                        for (int k = 0; k < j; k++) sacc -= L[i][k] * L[j][k];
                        if (i == j) {
                            ok = ok && (sacc > 0.0);
                            const double y = small_rsqrt(sacc);
                            I[j][j] = y;
                            L[j][j] = sacc * y;
                        } else {
                            L[i][j] = sacc * I[j][j];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 6; j++)
#pragma unroll
                    for (int i = j + 1; i < 6; i++) {
                        double sacc = 0.0;
#pragma unroll
                        for (int k = j; k < i; k++) sacc -= L[i][k] * I[k][j];
                        I[i][j] = sacc * I[i][i];
                    }
                if (!ok) scal[SC_LIN_STATUS] = 1.0;
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        M[(k0 + i) * ld + k0 + j] = j <= i ? L[i][j] : 0.0;
                        Li[kb * 36 + i * 6 + j] = j <= i ? I[i][j] : 0.0;
                    }
            }
        }
        __syncthreads();
        // panel: rows below the diagonal block, A_ik <- A_ik L_kk^-T
        const int rows_below = n - k0 - 6;
        if (tid < rows_below) {
            double* a = M + (size_t)(k0 + 6 + tid) * ld + k0;
            const double* I = Li + kb * 36;
            const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
            a[0] = a0 * I[0];
            a[1] = a0 * I[6] + a1 * I[7];
            a[2] = a0 * I[12] + a1 * I[13] + a2 * I[14];
            a[3] = a0 * I[18] + a1 * I[19] + a2 * I[20] + a3 * I[21];
            a[4] = a0 * I[24] + a1 * I[25] + a2 * I[26] + a3 * I[27] + a4 * I[28];
            a[5] = a0 * I[30] + a1 * I[31] + a2 * I[32] + a3 * I[33] + a4 * I[34] + a5 * I[35];
        }
        __syncthreads();
        // trailing update of the lower triangle: A_ij -= L_ik L_jk^T, one lane per (row i, block column jb <= ib)
        const int nbr = nb - 1 - kb;                       // block rows below
        const int n_items = rows_below * nbr;              // (row, jb) with jb <= row's block filtered below
        for (int item = tid; item < n_items; item += kSmallThreads) {
            const int jb = item / rows_below, ri = item - jb * rows_below;   // ri: row below the pivot block
            if (jb * 6 > ri) continue;                     // only the lower triangle (by block)
            const double* li = M + (size_t)(k0 + 6 + ri) * ld + k0;
            const double l0 = li[0], l1 = li[1], l2 = li[2], l3 = li[3], l4 = li[4], l5 = li[5];
            double* tgt = M + (size_t)(k0 + 6 + ri) * ld + k0 + 6 + jb * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double* lj = M + (size_t)(k0 + 6 + jb * 6 + c) * ld + k0;
                tgt[c] -= l0 * lj[0] + l1 * lj[1] + l2 * lj[2] + l3 * lj[3] + l4 * lj[4] + l5 * lj[5];
            }
        }
        __syncthreads();
    }

    // forward: y_k = L_kk^-1 v_k, then v_i -= L_ik y_k for the rows below
    for (int kb = 0; kb < nb; kb++) {
        const int k0 = 6 * kb;
        double yk = 0.0;
        if (tid < 6) {
            const double* I = Li + kb * 36 + tid * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) yk += I[c] * v[k0 + c];
        }
        __syncthreads();
        if (tid < 6) v[k0 + tid] = yk;
        __syncthreads();
        const int row = k0 + 6 + tid;
        if (row < n) {
            const double* l = M + (size_t)row * ld + k0;
            v[row] -= l[0] * v[k0] + l[1] * v[k0 + 1] + l[2] * v[k0 + 2] + l[3] * v[k0 + 3] + l[4] * v[k0 + 4] + l[5] * v[k0 + 5];
        }
        __syncthreads();
    }
    // backward: x_k = L_kk^-T y_k, then v_i -= L_ki^T x_k for the rows above
    for (int kb = nb - 1; kb >= 0; kb--) {
        const int k0 = 6 * kb;
        double xk = 0.0;
        if (tid < 6) {
            const double* I = Li + kb * 36;
#pragma unroll
            for (int c = 0; c < 6; c++) xk += I[c * 6 + tid] * v[k0 + c];
        }
        __syncthreads();
        if (tid < 6) v[k0 + tid] = xk;
        __syncthreads();
        if (tid < k0) {
            const double* l = M + (size_t)k0 * ld + tid;   // column tid of the block row k
            v[tid] -= l[0] * v[k0] + l[ld] * v[k0 + 1] + l[2 * ld] * v[k0 + 2] + l[3 * ld] * v[k0 + 3] + l[4 * ld] * v[k0 + 4] +
                      l[5 * ld] * v[k0 + 5];
        }
        __syncthreads();
    }
    if (tid < n) x[tid] = v[tid];
}


// ---- dense systems up to 1 260 unknowns: explicit inverse by block Gauss-Jordan (crsolve.hip) + one round of refinement -------
// The blocked right-looking Cholesky below pays 3 launches per 32 columns with a single-workgroup diagonal step (34 us) and a
// single-workgroup triangular solve behind it (290 us at 594 unknowns: BASELINE.json configs[1] took 1.25 ms per solve).  With
// the matrix padded to tiles of 60 and inverted explicitly on the f64 matrix cores (3 launches per 60 columns, every one a grid
// of products) the solve is two dense products and - as the one-workgroup solvers do - one refinement against the block-sparse
// S itself, which takes the explicit inverse's rounding out of the result.
__global__ __launch_bounds__(256) void dense60_fill_kernel(const uint32_t n, const uint32_t ncp, double* __restrict__ dense)
{
    // zero, with the identity on the padding
    const uint32_t row = blockIdx.x;
    for (uint32_t col = threadIdx.x; col < ncp; col += 256) dense[(size_t)row * ncp + col] = (row >= n && row == col) ? 1.0 : 0.0;
}

__global__ __launch_bounds__(64) void dense60_scatter_kernel(const double* __restrict__ blocks, const uint32_t* __restrict__ blk_row,
                                                             const uint32_t* __restrict__ blk_col, const uint32_t ncp, double* __restrict__ dense)
{
    const uint32_t blk = blockIdx.x;
    const int t = threadIdx.x;
    if (t >= 36) return;
    const uint32_t i = 6 * blk_row[blk] + t / 6, j = 6 * blk_col[blk] + t % 6;
    const double v = blocks[36 * (size_t)blk + t];
    dense[(size_t)i * ncp + j] = v;
    dense[(size_t)j * ncp + i] = v;
}

// out = base + Ainv v (one wave per row, lane-strided sum in a fixed order); MODE 1: v = b - S x first (written to r)
__global__ __launch_bounds__(256) void dense60_apply_kernel(const double* __restrict__ Ainv, const uint32_t ncp, const uint32_t n,
                                                            const double* __restrict__ v, const double* __restrict__ base, double* __restrict__ out)
{
    const uint32_t row = blockIdx.x * 4 + threadIdx.x / 64;
    const int lane = threadIdx.x % 64;
    if (row >= n) return;
    double sum = 0.0;
    const double* a = Ainv + (size_t)row * ncp;
    for (uint32_t j = lane; j < n; j += 64) sum += a[j] * v[j];
    sum = wave_sum(sum);
    if (lane == 0) out[row] = (base ? base[row] : 0.0) + sum;
}

__global__ __launch_bounds__(256) void bsr_residual_kernel(const BsrView A, const double* __restrict__ b, const double* __restrict__ x,
                                                           double* __restrict__ r)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n_rows * 6) return;
    const uint32_t f = i / 6, a = i % 6;
    double s = b[i];
    for (uint32_t e = A.row_ptr[f]; e < A.row_ptr[f + 1]; e++) {
        const double* B = A.blocks + 36 * (size_t)A.ent_blk[e];
        const double* xv = x + 6 * (size_t)A.ent_col[e];
        if (A.ent_trans[e]) {
#pragma unroll
            for (int c = 0; c < 6; c++) s -= B[c * 6 + a] * xv[c];
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) s -= B[a * 6 + c] * xv[c];
        }
    }
    r[i] = s;
}

}  // namespace

bool dense_small_fits(uint32_t n_rows) { return n_rows >= 1 && n_rows <= (uint32_t)kDenseSmallRowsMax; }

void launch_dense_small_solve(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col,
                              const double* b, double* x, double* scal)
{
    if (!A.n_rows) return;
    const size_t n = 6 * (size_t)A.n_rows, ld = n | 1;
    const size_t lds = sizeof(double) * (n * ld + (size_t)A.n_rows * 36 + n);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_small_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dense_small_solve_kernel, dim3(1), dim3(kSmallThreads), lds, s, A, n_blocks, blk_row, blk_col, b, x, scal);
}

void launch_bsr_to_dense(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row,
                         const uint32_t* blk_col, double* dense)
{
    const uint32_t n = A.n_rows * 6;
    if (!n) return;
    (void)hipMemsetAsync(dense, 0, sizeof(double) * (size_t)n * n, s);
    hipLaunchKernelGGL(bsr_to_dense_kernel, dim3(n_blocks), dim3(64), 0, s, A.blocks, blk_row, blk_col, n, dense);
}

void launch_dense_cholesky_solve(hipStream_t s, uint32_t n, double* dense, const double* b, double* x, double* scal)
{
    if (!n) return;
    // workspace for the inverted diagonal blocks sits behind the matrix (caller allocates n*n + nblk*NB*NB)
    double* dinv = dense + (size_t)n * n;
    for (uint32_t k0 = 0; k0 < n; k0 += kNB) {
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s, dense, n, k0, dinv, scal);
        if (k0 + kNB < n) {
            const uint32_t rows = n - k0 - kNB;
            hipLaunchKernelGGL(chol_panel_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, dense, n, k0);
            const uint32_t tiles = (rows + kNB - 1) / kNB;
            hipLaunchKernelGGL(chol_update_kernel, dim3(tiles, tiles), dim3(256), 0, s, dense, n, k0);
        }
    }
    hipLaunchKernelGGL(chol_solve_kernel, dim3(1), dim3(1024), 0, s, dense, n, dinv, b, x);
}

bool dense_inverse_fits(uint32_t n_rows) { return n_rows >= 1 && n_rows * 6 <= 1260; }

size_t dense_inverse_count(uint32_t n_rows)
{
    const size_t ncp = ((size_t)n_rows * 6 + 59) / 60 * 60;
    return ncp * ncp + 2 * 3600 + 3 * ncp;
}

void launch_dense_inverse_solve(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row, const uint32_t* blk_col,
                                double* work, const double* b, double* x, double* scal)
{
    const uint32_t n = A.n_rows * 6;
    if (!n) return;
    const uint32_t ncp = (n + 59) / 60 * 60;
    double* dense = work;
    double* ebuf = dense + (size_t)ncp * ncp;
    double* x0 = ebuf + 2 * 3600;
    double* r = x0 + ncp;
    hipLaunchKernelGGL(dense60_fill_kernel, dim3(ncp), dim3(256), 0, s, n, ncp, dense);
    hipLaunchKernelGGL(dense60_scatter_kernel, dim3(n_blocks), dim3(64), 0, s, A.blocks, blk_row, blk_col, ncp, dense);
    launch_dense_spd_inverse60(s, dense, (int)(ncp / 60), ebuf, scal + SC_LIN_STATUS);
    hipLaunchKernelGGL(dense60_apply_kernel, dim3((n + 3) / 4), dim3(256), 0, s, dense, ncp, n, b, nullptr, x0);
    hipLaunchKernelGGL(bsr_residual_kernel, dim3((n + 255) / 256), dim3(256), 0, s, A, b, x0, r);
    hipLaunchKernelGGL(dense60_apply_kernel, dim3((n + 3) / 4), dim3(256), 0, s, dense, ncp, n, r, x0, x);
}

}  // namespace soslam
