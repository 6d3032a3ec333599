// linsolve.hip - reduced camera system solvers for gfx950: block-Jacobi PCG on the block-sparse S and a
// blocked right-looking dense Cholesky.  The systems are small (<= a few thousand unknowns; 2.7 MB of
// blocks at BASELINE.json configs[2]), so both are latency-bound, not HBM-bound; they run as a handful of
// workgroups and keep the host out of the loop (no per-iteration synchronisation).
#include "linsolve.h"

#include "ba_kernels.h"

namespace soslam {

namespace {

constexpr int kWave = 64;
constexpr int kNB = 32;  // dense Cholesky panel width

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, kWave);
    return x;
}

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

}  // namespace soslam
