// bandsolve.hip - block-band Cholesky of the reduced camera matrix and PCG preconditioned by it (gfx950).
//
// A visual-odometry window couples camera i only with cameras i-bw..i+bw, so S is block-banded
// (bw = 9 blocks at BASELINE.json configs[1..2]).  The factorisation is a chain of n dependent block
// steps - latency-bound, not bandwidth-bound - so ONE workgroup runs it with the active (bw+2) block rows
// in an LDS ring: per step every lane redundantly factors the 6x6 pivot block in registers (no broadcast
// barrier), bw*6 lanes solve the panel rows, all lanes apply the trailing update, and the finished row is
// swapped for the next one of the band.  Two barriers per block step.
#include "linsolve.h"

#include "ba_kernels.h"

namespace soslam {

namespace {

constexpr int kWave = 64;
constexpr int kSolveChunk = 8;   // block rows staged per LDS chunk in the triangular solves

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, kWave);
    return x;
}

__device__ __forceinline__ double block_sum(double v, double* red)
{
    v = wave_sum(v);
    __syncthreads();
    if (threadIdx.x % kWave == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    double t = 0.0;
    const int nw = blockDim.x / kWave;
    for (int w = 0; w < nw; w++) t += red[w];
    return t;
}

__global__ __launch_bounds__(64) void bsr_to_band_kernel(const double* __restrict__ blocks, const uint32_t* __restrict__ blk_row,
                                                         const uint32_t* __restrict__ blk_col, int bw, double* __restrict__ band)
{
    const uint32_t blk = blockIdx.x;
    const int t = threadIdx.x;
    if (t >= 36) return;
    const uint32_t i = blk_row[blk], j = blk_col[blk];   // i <= j: upper block; its transpose is lower block (j, i)
    const int d = (int)(j - i);
    if (d > bw) return;
    const int a = t / 6, b = t % 6;
    band[((size_t)j * (bw + 1) + d) * 36 + b * 6 + a] = blocks[36 * (size_t)blk + t];
}

// Cholesky factor L (lower) of a 6x6 SPD block and its inverse, fully in registers
__device__ __forceinline__ bool chol6(const double* __restrict__ A, double (&L)[6][6], double (&Li)[6][6])
{
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
                Li[j][j] = 1.0 / L[j][j];
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

__global__ __launch_bounds__(1024) void band_cholesky_kernel(double* __restrict__ band, const uint32_t n, const int bw,
                                                             double* __restrict__ dinv, double* __restrict__ scal)
{
    extern __shared__ double ring[];
    const int W = bw + 1, rowlen = W * 36, nslot = bw + 2;
    const int tid = threadIdx.x, NT = blockDim.x;
    const uint32_t pre = n < (uint32_t)nslot ? n : (uint32_t)nslot;
    for (uint32_t r = 0; r < pre; r++)
        for (int e = tid; e < rowlen; e += NT) ring[(r % nslot) * rowlen + e] = band[(size_t)r * rowlen + e];
    __syncthreads();

    for (uint32_t k = 0; k < n; k++) {
        double* rowk = ring + (k % nslot) * rowlen;
        double L[6][6], Li[6][6];
        const bool ok = chol6(rowk, L, Li);
        if (!ok && tid == 0) scal[SC_LIN_STATUS] = 1.0;
        const int nb = (int)min((uint32_t)bw, n - 1 - k);
        __syncthreads();  // every lane has read A_kk before it is overwritten below
        if (tid < nb * 6) {
            // panel: row r of L_ik = A_ik L_kk^-T
            const uint32_t i = k + 1 + tid / 6;
            double* a = ring + (i % nslot) * rowlen + (int)(i - k) * 36 + (tid % 6) * 6;
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
            rowk[e] = lv;
            dinv[36 * (size_t)k + e] = iv;
        }
        __syncthreads();
        // trailing update of the window's lower block triangle: A_ij -= L_ik L_jk^T, k < j <= i <= k+nb
        const int n_ent = nb * (nb + 1) / 2 * 36;
        for (int e = tid; e < n_ent; e += NT) {
            const int pair = e / 36, rc = e % 36, r = rc / 6, c = rc % 6;
            int ii = (int)((sqrtf(8.0f * (float)pair + 1.0f) - 1.0f) * 0.5f);
            while ((ii + 1) * (ii + 2) / 2 <= pair) ii++;
            while (ii * (ii + 1) / 2 > pair) ii--;
            const int jj = pair - ii * (ii + 1) / 2;
            const uint32_t i = k + 1 + ii, j = k + 1 + jj;
            double* ri = ring + (i % nslot) * rowlen;
            const double* lik = ri + (int)(i - k) * 36 + r * 6;
            const double* ljk = ring + (j % nslot) * rowlen + (int)(j - k) * 36 + c * 6;
            ri[(int)(i - j) * 36 + rc] -= lik[0] * ljk[0] + lik[1] * ljk[1] + lik[2] * ljk[2] + lik[3] * ljk[3] + lik[4] * ljk[4] + lik[5] * ljk[5];
        }
        // row k is final: store it and bring in the next row of the band (not touched by the update above)
        for (int e = tid; e < rowlen; e += NT) {
            band[(size_t)k * rowlen + e] = rowk[e];
            if (k + nslot < n) rowk[e] = band[(size_t)(k + nslot) * rowlen + e];
        }
        __syncthreads();
    }
}

// x = (L L^T)^-1 b.  x doubles as the intermediate y; lds holds one chunk of factor rows, their inverse
// diagonal blocks, a ring of the last bw solution blocks and the per-(d,c) partial sums.
__device__ void band_solve_device(const uint32_t n, const int bw, const double* __restrict__ band,
                                  const double* __restrict__ dinv, const double* __restrict__ b, double* __restrict__ x,
                                  double* lds)
{
    const int W = bw + 1, rowlen = W * 36;
    const int tid = threadIdx.x, NT = blockDim.x;
    double* rows = lds;                                        // (kSolveChunk + bw) * rowlen
    double* di = rows + (size_t)(kSolveChunk + bw) * rowlen;   // kSolveChunk * 36
    double* part = di + kSolveChunk * 36;                      // bw * 6
    double* sol = part + bw * 6;                               // (kSolveChunk + bw) * 6 : solution blocks of the window

    // ---- forward: L y = b, chunk of rows [k0, k0 + CH) needs y of rows k0-bw .. k0-1 --------------------
    for (uint32_t k0 = 0; k0 < n; k0 += kSolveChunk) {
        const int ch = (int)min((uint32_t)kSolveChunk, n - k0);
        __syncthreads();
        for (int e = tid; e < ch * rowlen; e += NT) rows[e] = band[(size_t)k0 * rowlen + e];
        for (int e = tid; e < ch * 36; e += NT) di[e] = dinv[36 * (size_t)k0 + e];
        // sol[0 .. bw) = y of rows k0-bw .. k0-1 (zero where negative), sol[bw + i] = b of row k0 + i
        for (int e = tid; e < (bw + ch) * 6; e += NT) {
            const long row = (long)k0 - bw + e / 6;
            sol[e] = row < 0 ? 0.0 : (row < (long)k0 ? x[6 * row + e % 6] : b[6 * row + e % 6]);
        }
        __syncthreads();
        for (int kk = 0; kk < ch; kk++) {
            if (tid < bw * 6) {
                const int d = 1 + tid / 6, c = tid % 6;
                const double* l = rows + kk * rowlen + d * 36 + c * 6;
                const double* y = sol + (bw + kk - d) * 6;
                part[tid] = l[0] * y[0] + l[1] * y[1] + l[2] * y[2] + l[3] * y[3] + l[4] * y[4] + l[5] * y[5];
            }
            __syncthreads();
            double ya = 0.0;
            if (tid < 6) {
                const double* M = di + kk * 36 + tid * 6;   // row tid of Linv (lower)
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double t = sol[(bw + kk) * 6 + c];
                    for (int d = 0; d < bw; d++) t -= part[d * 6 + c];
                    ya += M[c] * t;
                }
            }
            __syncthreads();
            if (tid < 6) sol[(bw + kk) * 6 + tid] = ya;
            __syncthreads();
        }
        for (int e = tid; e < ch * 6; e += NT) x[6 * (size_t)k0 + e] = sol[bw * 6 + e];
    }
    // ---- backward: L^T x = y, row k needs block d of rows k+1 .. k+bw ------------------------------------
    const uint32_t nchunk = (n + kSolveChunk - 1) / kSolveChunk;
    for (uint32_t cb = nchunk; cb-- > 0;) {
        const uint32_t k0 = cb * kSolveChunk;
        const int ch = (int)min((uint32_t)kSolveChunk, n - k0);
        const int nrow = (int)min((uint32_t)(ch + bw), n - k0);   // rows k0 .. k0+nrow-1 staged
        __syncthreads();
        for (int e = tid; e < nrow * rowlen; e += NT) rows[e] = band[(size_t)k0 * rowlen + e];
        for (int e = tid; e < ch * 36; e += NT) di[e] = dinv[36 * (size_t)k0 + e];
        // sol[i] = x (already solved) of row k0 + i for i >= ch, y of row k0 + i for i < ch
        for (int e = tid; e < (ch + bw) * 6; e += NT) {
            const uint32_t row = k0 + e / 6;
            sol[e] = row < n ? x[6 * (size_t)row + e % 6] : 0.0;
        }
        __syncthreads();
        for (int kk = ch - 1; kk >= 0; kk--) {
            if (tid < bw * 6) {
                const int d = 1 + tid / 6, m = tid % 6;
                double s = 0.0;
                if (kk + d < nrow) {
                    const double* l = rows + (kk + d) * rowlen + d * 36 + m;   // column m of L_{k+d,k}
                    const double* xv = sol + (kk + d) * 6;
                    s = l[0] * xv[0] + l[6] * xv[1] + l[12] * xv[2] + l[18] * xv[3] + l[24] * xv[4] + l[30] * xv[5];
                }
                part[tid] = s;
            }
            __syncthreads();
            double xa = 0.0;
            if (tid < 6) {
                const double* M = di + kk * 36;   // x_a = sum_c Linv[c][a] t[c]
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double t = sol[kk * 6 + c];
                    for (int d = 0; d < bw; d++) t -= part[d * 6 + c];
                    xa += M[c * 6 + tid] * t;
                }
            }
            __syncthreads();
            if (tid < 6) sol[kk * 6 + tid] = xa;
            __syncthreads();
        }
        for (int e = tid; e < ch * 6; e += NT) x[6 * (size_t)k0 + e] = sol[e];
    }
    __syncthreads();
}

__host__ __device__ inline size_t solve_lds_count(int bw)
{
    return (size_t)(kSolveChunk + bw) * (bw + 1) * 36 + kSolveChunk * 36 + (size_t)bw * 6 + (size_t)(kSolveChunk + bw) * 6;
}

__global__ __launch_bounds__(256) void band_solve_kernel(const uint32_t n, const int bw, const double* __restrict__ band,
                                                         const double* __restrict__ dinv, const double* __restrict__ b,
                                                         double* __restrict__ x)
{
    extern __shared__ double lds[];
    band_solve_device(n, bw, band, dinv, b, x, lds);
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

// PCG on S with M = band factor.  Convergence is tested right after the residual update, so when the band
// holds all of S the loop ends after ONE product with S and ONE preconditioner solve.
__global__ __launch_bounds__(256) void pcg_band_kernel(const BsrView A, const int bw, const double* __restrict__ band,
                                                       const double* __restrict__ dinv, const double* __restrict__ b,
                                                       double* __restrict__ x, double* __restrict__ r, double* __restrict__ work,
                                                       const double tol, const int max_iter, double* __restrict__ scal)
{
    extern __shared__ double lds[];
    __shared__ double red[4];
    const uint32_t n = A.n_rows * 6;
    double* p = work;
    double* z = work + n;
    double* q = work + 2 * (size_t)n;
    const int tid = threadIdx.x, NT = blockDim.x;

    double bb = 0.0;
    for (uint32_t i = tid; i < n; i += NT) {
        const double bi = b[i];
        x[i] = 0.0;
        r[i] = bi;
        bb += bi * bi;
    }
    bb = block_sum(bb, red);
    band_solve_device(A.n_rows, bw, band, dinv, r, z, lds);
    double rz = 0.0;
    for (uint32_t i = tid; i < n; i += NT) {
        const double zi = z[i];
        p[i] = zi;
        rz += r[i] * zi;
    }
    rz = block_sum(rz, red);
    const double stop2 = tol * tol * bb;
    double rr = bb;
    int it = 0;
    while (it < max_iter && rr > stop2 && rr > 0.0) {
        double pq = 0.0;
        for (uint32_t i = tid; i < n; i += NT) {
            const double qi = bsr_row_dot(A, i, p);
            q[i] = qi;
            pq += p[i] * qi;
        }
        pq = block_sum(pq, red);
        if (!(pq > 0.0)) { if (tid == 0) scal[SC_LIN_STATUS] = 2.0; break; }
        const double alpha = rz / pq;
        double rr_new = 0.0;
        for (uint32_t i = tid; i < n; i += NT) {
            x[i] += alpha * p[i];
            const double ri = r[i] - alpha * q[i];
            r[i] = ri;
            rr_new += ri * ri;
        }
        rr = block_sum(rr_new, red);
        it++;
        if (!(rr > stop2)) break;
        band_solve_device(A.n_rows, bw, band, dinv, r, z, lds);
        double rz_new = 0.0;
        for (uint32_t i = tid; i < n; i += NT) rz_new += r[i] * z[i];
        rz_new = block_sum(rz_new, red);
        const double beta = rz_new / rz;
        rz = rz_new;
        for (uint32_t i = tid; i < n; i += NT) p[i] = z[i] + beta * p[i];
        __syncthreads();
    }
    if (tid == 0) {
        scal[SC_LIN_ITERS] = (double)it;
        scal[SC_LIN_RESID] = bb > 0.0 ? sqrt(rr / bb) : 0.0;
    }
}

}  // namespace

size_t band_count(uint32_t n_rows, int bw) { return (size_t)n_rows * (bw + 1) * 36; }

void launch_bsr_to_band(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row,
                        const uint32_t* blk_col, int bw, double* band)
{
    if (!A.n_rows) return;
    (void)hipMemsetAsync(band, 0, sizeof(double) * band_count(A.n_rows, bw), s);
    hipLaunchKernelGGL(bsr_to_band_kernel, dim3(n_blocks), dim3(64), 0, s, A.blocks, blk_row, blk_col, bw, band);
}

void launch_band_cholesky(hipStream_t s, uint32_t n_rows, int bw, double* band, double* dinv, double* scal)
{
    if (!n_rows) return;
    const size_t lds = sizeof(double) * (size_t)(bw + 2) * (bw + 1) * 36;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_cholesky_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(band_cholesky_kernel, dim3(1), dim3(1024), lds, s, band, n_rows, bw, dinv, scal);
}

void launch_band_solve(hipStream_t s, uint32_t n_rows, int bw, const double* band, const double* dinv,
                       const double* b, double* x)
{
    if (!n_rows) return;
    const size_t lds = sizeof(double) * solve_lds_count(bw);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(band_solve_kernel, dim3(1), dim3(256), lds, s, n_rows, bw, band, dinv, b, x);
}

void launch_pcg_band(hipStream_t s, const BsrView& A, int bw, const double* band, const double* dinv,
                     const double* b, double* x, double* resid, double* work, double tol, int max_iter, double* scal)
{
    if (!A.n_rows) return;
    const size_t lds = sizeof(double) * solve_lds_count(bw);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pcg_band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(pcg_band_kernel, dim3(1), dim3(256), lds, s, A, bw, band, dinv, b, x, resid, work, tol, max_iter, scal);
}

}  // namespace soslam
