// bandsolve.hip - block-band Cholesky of the reduced camera matrix and PCG preconditioned by it (gfx950).
//
// A visual-odometry window couples camera i only with cameras i-bw..i+bw, so S is block-banded
// (bw = 9 blocks at BASELINE.json configs[1..2]).  Factorisation and triangular solves are chains of n
// dependent block steps: latency-bound, not bandwidth-bound.  Design for that:
//   * ONE workgroup factors; the active (bw+2) block rows live in an LDS ring; the next rows of the band
//     are fetched from HBM several steps ahead into a register queue, so no step waits on a global load;
//   * every lane factors the 6x6 pivot redundantly in registers (v_rsq_f64 + two Newton steps, no division),
//     so the pivot needs no broadcast barrier; two barriers per block step;
//   * the factor is written twice, row-oriented (band) and column-oriented (bandT), so that BOTH
//     triangular solves run right-looking: after one block of the solution is known, all dependent
//     right-hand-side blocks are updated independently (no per-step reduction);
//   * a solve is ONE wave (workgroup barriers cost nothing there) with its own register prefetch queue.
#include "linsolve.h"

#include "ba_kernels.h"

namespace soslam {

namespace {

constexpr int kWave = 64;
constexpr int kFactorThreads = 320;
constexpr int kFactorQueue = 3;   // band rows in flight from HBM during the factorisation
constexpr int kSolveQueue = 4;    // factor rows in flight during a triangular solve

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

// 1/sqrt(s) to full f64 precision: hardware estimate + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double s)
{
    double y = __builtin_amdgcn_rsq(s);
    y = y * (1.5 - 0.5 * s * y * y);
    y = y * (1.5 - 0.5 * s * y * y);
    return y;
}

// Cholesky factor L (lower) of a 6x6 SPD block and its inverse, fully in registers, division-free
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

__global__ __launch_bounds__(kFactorThreads) void band_cholesky_kernel(double* __restrict__ band, double* __restrict__ bandT,
                                                                        const uint32_t n, const int bw,
                                                                        double* __restrict__ dinv, double* __restrict__ scal)
{
    extern __shared__ double ring[];
    constexpr int NT = kFactorThreads, Q = kFactorQueue;
    const int W = bw + 1, rowlen = W * 36, nslot = bw + 2;
    const int tid = threadIdx.x;
    const int e0 = tid, e1 = tid + NT;            // the (at most two) row elements this lane moves
    const bool has0 = e0 < rowlen, has1 = e1 < rowlen;

    const uint32_t pre_rows = n < (uint32_t)nslot ? n : (uint32_t)nslot;
    for (uint32_t r = 0; r < pre_rows; r++) {
        if (has0) ring[r * rowlen + e0] = band[(size_t)r * rowlen + e0];
        if (has1) ring[r * rowlen + e1] = band[(size_t)r * rowlen + e1];
    }
    // register queue: rows nslot .. nslot+Q-1
    double q0[Q], q1[Q];
#pragma unroll
    for (int j = 0; j < Q; j++) {
        const uint32_t row = (uint32_t)nslot + j;
        q0[j] = (has0 && row < n) ? band[(size_t)row * rowlen + e0] : 0.0;
        q1[j] = (has1 && row < n) ? band[(size_t)row * rowlen + e1] : 0.0;
    }
    // trailing-update items of this lane: (pair of window rows, row r of the 6x6 block); pairs are ordered by
    // their larger row so that the items of a shorter window (tail of the matrix) are a prefix
    constexpr int kMaxItems = (kBandMax * (kBandMax + 1) / 2 * 6 + NT - 1) / NT;
    int it_ii[kMaxItems], it_jj[kMaxItems], it_r[kMaxItems];
#pragma unroll
    for (int s = 0; s < kMaxItems; s++) {
        const int item = tid + s * NT, pair = item / 6;
        int ii = (int)((sqrtf(8.0f * (float)pair + 1.0f) - 1.0f) * 0.5f);
        while ((ii + 1) * (ii + 2) / 2 <= pair) ii++;
        while (ii * (ii + 1) / 2 > pair) ii--;
        it_ii[s] = ii; it_jj[s] = pair - ii * (ii + 1) / 2; it_r[s] = item % 6;
    }
    __syncthreads();

    int slot_k = 0;   // k % nslot
    for (uint32_t k = 0; k < n; k++) {
        double* rowk = ring + slot_k * rowlen;
        double L[6][6], Li[6][6];
        const bool ok = chol6(rowk, L, Li);
        if (!ok && tid == 0) scal[SC_LIN_STATUS] = 1.0;
        const int nb = (int)min((uint32_t)bw, n - 1 - k);
        __syncthreads();  // every lane has read A_kk before it is overwritten below
        if (tid < nb * 6) {
            // panel: row r of L_ik = A_ik L_kk^-T; also stored column-oriented for the forward solve
            const int d = 1 + tid / 6, r = tid % 6;
            int slot_i = slot_k + d;
            if (slot_i >= nslot) slot_i -= nslot;
            double* a = ring + slot_i * rowlen + d * 36 + r * 6;
            const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
            const double x0 = a0 * Li[0][0];
            const double x1 = a0 * Li[1][0] + a1 * Li[1][1];
            const double x2 = a0 * Li[2][0] + a1 * Li[2][1] + a2 * Li[2][2];
            const double x3 = a0 * Li[3][0] + a1 * Li[3][1] + a2 * Li[3][2] + a3 * Li[3][3];
            const double x4 = a0 * Li[4][0] + a1 * Li[4][1] + a2 * Li[4][2] + a3 * Li[4][3] + a4 * Li[4][4];
            const double x5 = a0 * Li[5][0] + a1 * Li[5][1] + a2 * Li[5][2] + a3 * Li[5][3] + a4 * Li[5][4] + a5 * Li[5][5];
            a[0] = x0; a[1] = x1; a[2] = x2; a[3] = x3; a[4] = x4; a[5] = x5;
            double2* o = reinterpret_cast<double2*>(bandT + ((size_t)k * W + d) * 36 + r * 6);
            o[0] = make_double2(x0, x1); o[1] = make_double2(x2, x3); o[2] = make_double2(x4, x5);
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
        const int n_items = nb * (nb + 1) / 2 * 6;
#pragma unroll
        for (int s = 0; s < kMaxItems; s++) {
            if (tid + s * NT < n_items) {
                int slot_i = slot_k + 1 + it_ii[s], slot_j = slot_k + 1 + it_jj[s];
                if (slot_i >= nslot) slot_i -= nslot;
                if (slot_j >= nslot) slot_j -= nslot;
                double* ri = ring + slot_i * rowlen;
                const double* lik = ri + (it_ii[s] + 1) * 36 + it_r[s] * 6;
                const double* ljk = ring + slot_j * rowlen + (it_jj[s] + 1) * 36;
                double* tgt = ri + (it_ii[s] - it_jj[s]) * 36 + it_r[s] * 6;
                const double l0 = lik[0], l1 = lik[1], l2 = lik[2], l3 = lik[3], l4 = lik[4], l5 = lik[5];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const double* lj = ljk + c * 6;
                    tgt[c] -= l0 * lj[0] + l1 * lj[1] + l2 * lj[2] + l3 * lj[3] + l4 * lj[4] + l5 * lj[5];
                }
            }
        }
        // row k is final: store it, give its slot to the oldest row of the register queue, refill the queue
        const uint32_t next = k + (uint32_t)nslot + Q;
        if (has0) {
            band[(size_t)k * rowlen + e0] = rowk[e0];
            rowk[e0] = q0[0];
        }
        if (has1) {
            band[(size_t)k * rowlen + e1] = rowk[e1];
            rowk[e1] = q1[0];
        }
#pragma unroll
        for (int j = 0; j + 1 < Q; j++) { q0[j] = q0[j + 1]; q1[j] = q1[j + 1]; }
        q0[Q - 1] = (has0 && next < n) ? band[(size_t)next * rowlen + e0] : 0.0;
        q1[Q - 1] = (has1 && next < n) ? band[(size_t)next * rowlen + e1] : 0.0;
        slot_k = slot_k + 1 == nslot ? 0 : slot_k + 1;
        __syncthreads();
    }
}

// ---- triangular solves: one wave, right-looking ------------------------------------------------------
//
// forward  L y = b   : y_k = Linv_kk t_k, then t_{k+d} -= L_{k+d,k} y_k (bandT row k, blocks d = 1..bw)
// backward L^T x = y : x_k = Linv_kk^T t_k, then t_{k-d} -= L_{k,k-d}^T x_k (band row k, blocks d = 1..bw)
// t lives in an LDS ring of bw+1 blocks; lane e owns update item (d = 1 + e/6, c = e%6), and a second item
// e + 64 when bw > 10.  ITEMS = 1 or 2.
template <int ITEMS>
__device__ void band_solve_wave(const uint32_t n, const int bw, const double* __restrict__ band,
                                const double* __restrict__ bandT, const double* __restrict__ dinv,
                                const double* __restrict__ b, double* __restrict__ x, double* tring, double* yk)
{
    constexpr int PF = kSolveQueue;
    const int W = bw + 1;
    const int lane = threadIdx.x;
    const int n_upd = bw * 6;

    // ================= forward =================
    {
        double lt[PF][ITEMS][6], di[PF][6], bq[PF];
        auto load_row = [&](uint32_t row, double (&l)[ITEMS][6], double (&dv)[6], double& bv) {
#pragma unroll
            for (int s = 0; s < ITEMS; s++) {
                const int e = lane + s * kWave;
                if (e < n_upd && row + 1 + (uint32_t)(e / 6) < n) {   // block L_{row+d,row} exists
                    const double2* p = reinterpret_cast<const double2*>(bandT + ((size_t)row * W + 1 + e / 6) * 36 + (e % 6) * 6);
                    const double2 v0 = p[0], v1 = p[1], v2 = p[2];
                    l[s][0] = v0.x; l[s][1] = v0.y; l[s][2] = v1.x; l[s][3] = v1.y; l[s][4] = v2.x; l[s][5] = v2.y;
                } else {
#pragma unroll
                    for (int m = 0; m < 6; m++) l[s][m] = 0.0;
                }
            }
            if (lane < 6 && row < n) {
                const double2* p = reinterpret_cast<const double2*>(dinv + 36 * (size_t)row + lane * 6);
                const double2 v0 = p[0], v1 = p[1], v2 = p[2];
                dv[0] = v0.x; dv[1] = v0.y; dv[2] = v1.x; dv[3] = v1.y; dv[4] = v2.x; dv[5] = v2.y;
                const uint32_t br = row + (uint32_t)W;   // the row that enters the ring when `row` leaves it
                bv = br < n ? b[6 * (size_t)br + lane] : 0.0;
            } else {
#pragma unroll
                for (int m = 0; m < 6; m++) dv[m] = 0.0;
                bv = 0.0;
            }
        };
#pragma unroll
        for (int j = 0; j < PF; j++) load_row((uint32_t)j, lt[j], di[j], bq[j]);
        for (int e = lane; e < W * 6; e += kWave) tring[e] = (uint32_t)(e / 6) < n ? b[e] : 0.0;
        __syncthreads();
        int slot_k = 0;
        for (uint32_t k = 0; k < n; k++) {
            if (lane < 6) {
                const double* t = tring + slot_k * 6;
                const double ya = di[0][0] * t[0] + di[0][1] * t[1] + di[0][2] * t[2] + di[0][3] * t[3] + di[0][4] * t[4] + di[0][5] * t[5];
                yk[lane] = ya;
                x[6 * (size_t)k + lane] = ya;
            }
            __syncthreads();
            const double y0 = yk[0], y1 = yk[1], y2 = yk[2], y3 = yk[3], y4 = yk[4], y5 = yk[5];
#pragma unroll
            for (int s = 0; s < ITEMS; s++) {
                const int e = lane + s * kWave;
                if (e < n_upd) {
                    int slot = slot_k + 1 + e / 6;
                    if (slot >= W) slot -= W;
                    tring[slot * 6 + e % 6] -= lt[0][s][0] * y0 + lt[0][s][1] * y1 + lt[0][s][2] * y2 + lt[0][s][3] * y3 + lt[0][s][4] * y4 + lt[0][s][5] * y5;
                }
            }
            if (lane < 6) tring[slot_k * 6 + lane] = bq[0];   // row k + W takes the freed slot
#pragma unroll
            for (int j = 0; j + 1 < PF; j++) {
#pragma unroll
                for (int s = 0; s < ITEMS; s++)
#pragma unroll
                    for (int m = 0; m < 6; m++) lt[j][s][m] = lt[j + 1][s][m];
#pragma unroll
                for (int m = 0; m < 6; m++) di[j][m] = di[j + 1][m];
                bq[j] = bq[j + 1];
            }
            load_row(k + PF, lt[PF - 1], di[PF - 1], bq[PF - 1]);
            slot_k = slot_k + 1 == W ? 0 : slot_k + 1;
            __syncthreads();
        }
    }
    // ================= backward =================
    {
        double lr[PF][ITEMS][6], di[PF][6], yq[PF];
        // row index runs downwards: step j handles k = n-1-j
        auto load_row = [&](long row, double (&l)[ITEMS][6], double (&dv)[6], double& yv) {
#pragma unroll
            for (int s = 0; s < ITEMS; s++) {
                const int e = lane + s * kWave;
                if (e < n_upd && row >= 0) {
                    // column m = e%6 of L_{row,row-d}: elements [c][m], c = 0..5
                    const double* p = band + ((size_t)row * W + 1 + e / 6) * 36 + (e % 6);
#pragma unroll
                    for (int c = 0; c < 6; c++) l[s][c] = p[c * 6];
                } else {
#pragma unroll
                    for (int c = 0; c < 6; c++) l[s][c] = 0.0;
                }
            }
            if (lane < 6 && row >= 0) {
                const double* p = dinv + 36 * (size_t)row + lane;   // column `lane` of Linv
#pragma unroll
                for (int c = 0; c < 6; c++) dv[c] = p[c * 6];
                const long yr = row - W;
                yv = yr >= 0 ? x[6 * (size_t)yr + lane] : 0.0;
            } else {
#pragma unroll
                for (int c = 0; c < 6; c++) dv[c] = 0.0;
                yv = 0.0;
            }
        };
        __syncthreads();   // forward's writes of x are complete (same wave, but keep the model clean)
#pragma unroll
        for (int j = 0; j < PF; j++) load_row((long)n - 1 - j, lr[j], di[j], yq[j]);
        // ring slot s holds row (n-1) - s for s = 0..bw at the start
        for (int e = lane; e < W * 6; e += kWave) {
            const long row = (long)n - 1 - e / 6;
            tring[e] = row >= 0 ? x[6 * (size_t)row + e % 6] : 0.0;
        }
        __syncthreads();
        int slot_k = 0;
        for (long k = (long)n - 1; k >= 0; k--) {
            if (lane < 6) {
                const double* t = tring + slot_k * 6;
                const double xa = di[0][0] * t[0] + di[0][1] * t[1] + di[0][2] * t[2] + di[0][3] * t[3] + di[0][4] * t[4] + di[0][5] * t[5];
                yk[lane] = xa;
                x[6 * (size_t)k + lane] = xa;
            }
            __syncthreads();
            const double x0 = yk[0], x1 = yk[1], x2 = yk[2], x3 = yk[3], x4 = yk[4], x5 = yk[5];
#pragma unroll
            for (int s = 0; s < ITEMS; s++) {
                const int e = lane + s * kWave;
                if (e < n_upd) {
                    int slot = slot_k + 1 + e / 6;
                    if (slot >= W) slot -= W;
                    tring[slot * 6 + e % 6] -= lr[0][s][0] * x0 + lr[0][s][1] * x1 + lr[0][s][2] * x2 + lr[0][s][3] * x3 + lr[0][s][4] * x4 + lr[0][s][5] * x5;
                }
            }
            if (lane < 6) tring[slot_k * 6 + lane] = yq[0];   // row k - W takes the freed slot
#pragma unroll
            for (int j = 0; j + 1 < PF; j++) {
#pragma unroll
                for (int s = 0; s < ITEMS; s++)
#pragma unroll
                    for (int m = 0; m < 6; m++) lr[j][s][m] = lr[j + 1][s][m];
#pragma unroll
                for (int m = 0; m < 6; m++) di[j][m] = di[j + 1][m];
                yq[j] = yq[j + 1];
            }
            load_row(k - PF, lr[PF - 1], di[PF - 1], yq[PF - 1]);
            slot_k = slot_k + 1 == W ? 0 : slot_k + 1;
            __syncthreads();
        }
    }
}

// PCG state shared between the round kernels (device f64 slots behind the work vectors)
enum { PS_RZ = 0, PS_BB = 1, PS_DONE = 2, PS_ROUNDS = 3, PS_PQ = 4, PS_COUNT = 8 };

// mode 0: plain solve x = M^-1 b.
// mode 1: PCG round head: z = M^-1 r (z = x argument), then p = z + beta p with beta from r.z (one wave, n small)
__global__ __launch_bounds__(64) void band_solve_kernel(const uint32_t n, const int bw, const double* __restrict__ band,
                                                        const double* __restrict__ bandT, const double* __restrict__ dinv,
                                                        const double* __restrict__ b, double* __restrict__ x, const int mode,
                                                        double* __restrict__ p, double* __restrict__ state)
{
    __shared__ double tring[(kBandMax + 1) * 6];
    __shared__ double yk[6];
    if (mode == 1 && state[PS_DONE] != 0.0) return;
    if (bw > 10) band_solve_wave<2>(n, bw, band, bandT, dinv, b, x, tring, yk);
    else band_solve_wave<1>(n, bw, band, bandT, dinv, b, x, tring, yk);
    if (mode == 1) {
        __syncthreads();
        const int lane = threadIdx.x;
        double rz = 0.0;
        for (uint32_t i = lane; i < n * 6; i += kWave) rz += b[i] * x[i];
        rz = wave_sum(rz);
        rz = __shfl(rz, 0, kWave);
        const double rz_old = state[PS_RZ];
        const double beta = state[PS_ROUNDS] == 0.0 ? 0.0 : rz / rz_old;
        for (uint32_t i = lane; i < n * 6; i += kWave) p[i] = x[i] + beta * p[i];
        __syncthreads();
        if (lane == 0) state[PS_RZ] = rz;
    }
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

// x = 0, r = b, |b|^2, state reset (one workgroup)
__global__ __launch_bounds__(1024) void pcg_band_init_kernel(uint32_t n, const double* __restrict__ b, double* __restrict__ x,
                                                             double* __restrict__ r, double* __restrict__ state)
{
    __shared__ double red[16];
    double bb = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const double bi = b[i];
        x[i] = 0.0;
        r[i] = bi;
        bb += bi * bi;
    }
    bb = block_sum(bb, red);
    if (threadIdx.x == 0) {
        state[PS_RZ] = 0.0; state[PS_BB] = bb; state[PS_DONE] = bb > 0.0 ? 0.0 : 1.0; state[PS_ROUNDS] = 0.0; state[PS_PQ] = 0.0;
    }
}

// q = S p over many workgroups, per-workgroup partial p.q
__global__ __launch_bounds__(256) void pcg_band_matvec_kernel(const BsrView A, const double* __restrict__ p, double* __restrict__ q,
                                                              double* __restrict__ part, const double* __restrict__ state)
{
    __shared__ double red[4];
    if (state[PS_DONE] != 0.0) return;
    const uint32_t n = A.n_rows * 6;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    double pq = 0.0;
    if (i < n) {
        const double qi = bsr_row_dot(A, i, p);
        q[i] = qi;
        pq = p[i] * qi;
    }
    pq = block_sum(pq, red);
    if (threadIdx.x == 0) part[blockIdx.x] = pq;
}

// alpha, x += alpha p, r -= alpha q, convergence test (one workgroup)
__global__ __launch_bounds__(1024) void pcg_band_update_kernel(uint32_t n, uint32_t n_part, const double* __restrict__ part,
                                                               const double* __restrict__ p, const double* __restrict__ q,
                                                               double* __restrict__ x, double* __restrict__ r,
                                                               double* __restrict__ state, const double tol, double* __restrict__ scal)
{
    __shared__ double red[16];
    if (state[PS_DONE] != 0.0) return;
    double pq = 0.0;
    for (uint32_t i = threadIdx.x; i < n_part; i += 1024) pq += part[i];
    pq = block_sum(pq, red);
    if (!(pq > 0.0)) {
        if (threadIdx.x == 0) { scal[SC_LIN_STATUS] = 2.0; state[PS_DONE] = 1.0; }
        return;
    }
    const double alpha = state[PS_RZ] / pq;
    double rr = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        x[i] += alpha * p[i];
        const double ri = r[i] - alpha * q[i];
        r[i] = ri;
        rr += ri * ri;
    }
    rr = block_sum(rr, red);
    if (threadIdx.x == 0) {
        const double bb = state[PS_BB];
        const double rounds = state[PS_ROUNDS] + 1.0;
        state[PS_ROUNDS] = rounds;
        scal[SC_LIN_ITERS] = rounds;
        scal[SC_LIN_RESID] = sqrt(rr / bb);
        if (!(rr > tol * tol * bb)) state[PS_DONE] = 1.0;
    }
}

}  // namespace

size_t band_count(uint32_t n_rows, int bw) { return (size_t)n_rows * (bw + 1) * 36; }
size_t pcg_band_work_count(uint32_t n_rows) { return (size_t)n_rows * 6 * 3 + PS_COUNT + (n_rows * 6 + 255) / 256 + 8; }

void launch_bsr_to_band(hipStream_t s, const BsrView& A, uint32_t n_blocks, const uint32_t* blk_row,
                        const uint32_t* blk_col, int bw, double* band)
{
    if (!A.n_rows) return;
    (void)hipMemsetAsync(band, 0, sizeof(double) * band_count(A.n_rows, bw), s);
    hipLaunchKernelGGL(bsr_to_band_kernel, dim3(n_blocks), dim3(64), 0, s, A.blocks, blk_row, blk_col, bw, band);
}

void launch_band_cholesky(hipStream_t s, uint32_t n_rows, int bw, double* band, double* bandT, double* dinv, double* scal)
{
    if (!n_rows) return;
    const size_t lds = sizeof(double) * (size_t)(bw + 2) * (bw + 1) * 36;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_cholesky_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(band_cholesky_kernel, dim3(1), dim3(kFactorThreads), lds, s, band, bandT, n_rows, bw, dinv, scal);
}

void launch_band_solve(hipStream_t s, uint32_t n_rows, int bw, const double* band, const double* bandT, const double* dinv,
                       const double* b, double* x)
{
    if (!n_rows) return;
    hipLaunchKernelGGL(band_solve_kernel, dim3(1), dim3(64), 0, s, n_rows, bw, band, bandT, dinv, b, x, 0, nullptr, nullptr);
}

void launch_pcg_band(hipStream_t s, const BsrView& A, int bw, const double* band, const double* bandT, const double* dinv,
                     const double* b, double* x, double* resid, double* work, double tol, int max_rounds, double* scal)
{
    if (!A.n_rows) return;
    const uint32_t n = A.n_rows * 6, n_wg = (n + 255) / 256;
    double* p = work;
    double* z = work + n;
    double* q = work + 2 * (size_t)n;
    double* state = work + 3 * (size_t)n;
    double* part = state + PS_COUNT;
    hipLaunchKernelGGL(pcg_band_init_kernel, dim3(1), dim3(1024), 0, s, n, b, x, resid, state);
    for (int round = 0; round < max_rounds; round++) {
        // every kernel of a round returns at once when the previous round converged (device-side flag)
        hipLaunchKernelGGL(band_solve_kernel, dim3(1), dim3(64), 0, s, A.n_rows, bw, band, bandT, dinv, resid, z, 1, p, state);
        hipLaunchKernelGGL(pcg_band_matvec_kernel, dim3(n_wg), dim3(256), 0, s, A, p, q, part, state);
        hipLaunchKernelGGL(pcg_band_update_kernel, dim3(1), dim3(1024), 0, s, n, n_wg, part, p, q, x, resid, state, tol, scal);
    }
}

}  // namespace soslam
