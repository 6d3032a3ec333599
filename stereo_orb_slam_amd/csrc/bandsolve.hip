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
#include "reduce.h"

namespace soslam {

namespace {

constexpr int kWave = 64;
constexpr int kFactorThreads = 320;

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

// e / d and e % d for small non-negative e and a RUNTIME divisor d (integer division has no hardware support and
// costs ~40 instructions; this is ~8): float reciprocal estimate + one correction step
struct FastDiv {
    int d;
    float inv;
    __host__ __device__ explicit FastDiv(int d_) : d(d_), inv(1.0f / (float)d_) {}
    __device__ __forceinline__ void divmod(int e, int& q, int& r) const
    {
        q = (int)((float)e * inv);
        r = e - q * d;
        if (r < 0) { q--; r += d; }
        if (r >= d) { q++; r -= d; }
    }
};

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

// Chunked band Cholesky.  LDS holds the rows [k0, k0 + C + bw] of the band (C pivot rows of the current chunk plus
// the bw rows they update); the C steps of a chunk touch LDS only - no global memory operation sits inside the
// dependent chain, so no step ever waits on s_waitcnt vmcnt.  At the chunk boundary the finished rows are written
// out (row-oriented band, column-oriented bandT, inverse diagonal blocks) and the next C rows are read into
// their ring slots: one exposed HBM round trip per C steps.
__global__ __launch_bounds__(kFactorThreads) void band_cholesky_kernel(double* __restrict__ band, double* __restrict__ bandT,
                                                                        const uint32_t n, const int bw, const int C,
                                                                        double* __restrict__ dinv, double* __restrict__ scal)
{
    extern __shared__ double lds[];
    constexpr int NT = kFactorThreads;
    constexpr int MAXQ = 18;                       // prefetch doubles per lane: C * rowlen / NT <= 18 (16*360/320, 8*576/320)
    const int W = bw + 1, rowlen = W * 36, R = W + C;
    double* ring = lds;                            // R * rowlen
    double* lkk = ring + (size_t)R * rowlen;       // C * 36 : factor of the chunk's diagonal blocks
    double* lki = lkk + C * 36;                    // C * 36 : their inverses
    const int tid = threadIdx.x;

    // rows 0 .. R-1
    const FastDiv by_row(rowlen);
    for (int e = tid; e < R * rowlen; e += NT) {
        int row, off;
        by_row.divmod(e, row, off);
        ring[e] = (uint32_t)row < n ? band[(size_t)e] : 0.0;
    }
    // trailing-update items of this lane (pair of window rows, row r of the 6x6 block), pairs ordered by larger row
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

    int slot_k0 = 0;   // ring slot of row k0 (k0 % R, tracked incrementally; R > C)
    for (uint32_t k0 = 0; k0 < n; k0 += (uint32_t)C) {
        const uint32_t next0 = k0 + (uint32_t)R;   // rows next0 .. next0 + C - 1 replace this chunk's rows at its end
        const int csteps = (int)min((uint32_t)C, n - k0);
        for (int kk = 0; kk < csteps; kk++) {
            const uint32_t k = k0 + kk;
            int slot_k = slot_k0 + kk;
            if (slot_k >= R) slot_k -= R;
            const double* rowk = ring + slot_k * rowlen;
            const int nb = (int)min((uint32_t)bw, n - 1 - k);
            if (tid < 128) {   // waves 0 and 1 hold the panel lanes and the writers of L_kk / its inverse
                double L[6][6], Li[6][6];
                const bool ok = chol6(rowk, L, Li);
                if (!ok && tid == 0) scal[SC_LIN_STATUS] = 1.0;
                if (tid < nb * 6) {
                    // panel: row r of L_ik = A_ik L_kk^-T
                    const int d = 1 + tid / 6, r = tid % 6;
                    int slot_i = slot_k + d;
                    if (slot_i >= R) slot_i -= R;
                    double* a = ring + slot_i * rowlen + d * 36 + r * 6;
                    const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
                    a[0] = a0 * Li[0][0];
                    a[1] = a0 * Li[1][0] + a1 * Li[1][1];
                    a[2] = a0 * Li[2][0] + a1 * Li[2][1] + a2 * Li[2][2];
                    a[3] = a0 * Li[3][0] + a1 * Li[3][1] + a2 * Li[3][2] + a3 * Li[3][3];
                    a[4] = a0 * Li[4][0] + a1 * Li[4][1] + a2 * Li[4][2] + a3 * Li[4][3] + a4 * Li[4][4];
                    a[5] = a0 * Li[5][0] + a1 * Li[5][1] + a2 * Li[5][2] + a3 * Li[5][3] + a4 * Li[5][4] + a5 * Li[5][5];
                } else if (tid >= 92 && tid < 128) {
                    const int e = tid - 92, r = e / 6, c = e % 6;
                    double lv = 0.0, iv = 0.0;
#pragma unroll
                    for (int rr = 0; rr < 6; rr++)
#pragma unroll
                        for (int cc = 0; cc < 6; cc++)
                            if (rr == r && cc == c && cc <= rr) { lv = L[rr][cc]; iv = Li[rr][cc]; }
                    lkk[kk * 36 + e] = lv;     // side buffers: A_kk itself stays intact, so no barrier before the panel
                    lki[kk * 36 + e] = iv;
                }
            }
            __syncthreads();
            // trailing update of the window's lower block triangle: A_ij -= L_ik L_jk^T, k < j <= i <= k+nb
            const int n_items = nb * (nb + 1) / 2 * 6;
#pragma unroll 1
            for (int s = 0; s < kMaxItems; s++) {
                if (tid + s * NT < n_items) {
                    int slot_i = slot_k + 1 + it_ii[s], slot_j = slot_k + 1 + it_jj[s];
                    if (slot_i >= R) slot_i -= R;
                    if (slot_j >= R) slot_j -= R;
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
            __syncthreads();
        }
        // ---- chunk boundary: write the finished rows out, then hand their slots to the prefetched rows ----
#pragma unroll
        for (int s = 0; s < MAXQ; s++) {
            int kk, off;
            by_row.divmod(tid + s * NT, kk, off);
            const int d = off / 36, rc = off % 36;
            if (tid + s * NT < csteps * rowlen) {
                const uint32_t k = k0 + kk;
                int slot = slot_k0 + kk;
                if (slot >= R) slot -= R;
                // row-oriented factor row k: block 0 = L_kk, blocks d >= 1 = L_{k,k-d} (final since step k-d)
                band[(size_t)k * rowlen + off] = d == 0 ? lkk[kk * 36 + rc] : ring[slot * rowlen + off];
                // column-oriented: bandT[k][d] = L_{k+d,k}, which sits in row k+d (still resident), block d
                int slot_d = slot + d;
                if (slot_d >= R) slot_d -= R;
                if (d >= 1 && k + d < n) bandT[(size_t)k * rowlen + off] = ring[slot_d * rowlen + off];
            }
        }
        for (int e = tid; e < csteps * 36; e += NT) dinv[36 * (size_t)k0 + e] = lki[e];
        __syncthreads();
        // one exposed HBM round trip per chunk of C steps (amortised: ~0.1 us per step)
#pragma unroll
        for (int s = 0; s < MAXQ; s++) {
            const int e = tid + s * NT;
            if (e < C * rowlen) {
                int row, off;
                by_row.divmod(e, row, off);
                int slot = slot_k0 + row;           // row k0 + R + row takes the slot of row k0 + row
                if (slot >= R) slot -= R;
                ring[slot * rowlen + off] = next0 + (uint32_t)row < n ? band[(size_t)next0 * rowlen + e] : 0.0;
            }
        }
        slot_k0 += C;
        if (slot_k0 >= R) slot_k0 -= R;
        __syncthreads();
    }
}

// ---- triangular solves: one wave, right-looking, chunked through LDS -----------------------------------
//
// forward  L y = b   : y_k = Linv_kk t_k, then t_{k+d} -= L_{k+d,k} y_k (bandT row k, blocks d = 1..bw)
// backward L^T x = y : x_k = Linv_kk^T t_k, then t_{k-d} -= L_{k,k-d}^T x_k (band row k, blocks d = 1..bw)
// t lives in an LDS ring of bw+1 blocks.  A chunk of C factor rows (+ their inverse diagonal blocks and the
// right-hand-side blocks that enter the ring) is staged in LDS; the next chunk is in flight into registers
// meanwhile, so the C dependent steps of a chunk run without any global memory operation.
template <int C, int IPR>   // C factor rows per chunk, IPR = ceil(max row length / 64) staged doubles per lane and row
__device__ void band_solve_wave(const uint32_t n, const int bw, const double* __restrict__ band,
                                const double* __restrict__ bandT, const double* __restrict__ dinv,
                                const double* __restrict__ b, double* __restrict__ x, double* lds)
{
    constexpr int NDI = (C * 36 + kWave - 1) / kWave;   // staged doubles per lane of the inverse diagonal blocks
    static_assert(C * 6 <= kWave, "one lane per entering right-hand-side value");
    const int W = bw + 1, rowlen = W * 36;
    const int lane = threadIdx.x;
    const int n_upd = bw * 6;
    double* rows = lds;                         // C * rowlen : factor rows of the chunk
    double* di = rows + (size_t)C * rowlen;     // C * 36
    double* rin = di + C * 36;                  // C * 6 : blocks entering the ring during the chunk
    double* xout = rin + C * 6;                 // C * 6 : solution blocks of the chunk
    double* tring = xout + C * 6;               // W * 6
    double* yk = tring + W * 6;                 // 6

    for (int dir = 0; dir < 2; dir++) {
        const bool fwd = dir == 0;
        const double* fac = fwd ? bandT : band;
        const double* rhs_in = fwd ? b : x;     // backward reads the y the forward pass left in x
        const uint32_t n_chunk = (n + (uint32_t)C - 1) / (uint32_t)C;
        // chunk c covers steps s = c*C .. c*C + C - 1; step s handles row k = s (forward) or n-1-s (backward)
        auto row_of = [&](long s) -> long { return fwd ? s : (long)n - 1 - s; };
        double qr[C][IPR], qd[NDI], qi;
        auto fetch = [&](uint32_t c) {
#pragma unroll
            for (int r = 0; r < C; r++) {
                const long k = row_of((long)c * C + r);
                const bool live = k >= 0 && k < (long)n;
#pragma unroll
                for (int i = 0; i < IPR; i++) {
                    const int e = lane + i * kWave;
                    qr[r][i] = (live && e < rowlen) ? fac[(size_t)k * rowlen + e] : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < NDI; i++) {
                const int f = lane + i * kWave;
                const long k = row_of((long)c * C + f / 36);
                qd[i] = (f < C * 36 && k >= 0 && k < (long)n) ? dinv[36 * (size_t)k + f % 36] : 0.0;
            }
            {
                const long s = (long)c * C + lane / 6 + W;               // the step whose row enters the ring
                const long k = row_of(s);
                qi = (lane < C * 6 && s < (long)n && k >= 0 && k < (long)n) ? rhs_in[6 * (size_t)k + lane % 6] : 0.0;
            }
        };
        __syncthreads();
        // ring slot j holds the row of step j for j = 0..bw
        for (int e = lane; e < W * 6; e += kWave) {
            const long k = row_of(e / 6);
            tring[e] = (e / 6 < (long)n && k >= 0 && k < (long)n) ? rhs_in[6 * (size_t)k + e % 6] : 0.0;
        }
        fetch(0);
        int slot_k = 0;
        for (uint32_t c = 0; c < n_chunk; c++) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < C; r++)
#pragma unroll
                for (int i = 0; i < IPR; i++) {
                    const int e = lane + i * kWave;
                    if (e < rowlen) rows[r * rowlen + e] = qr[r][i];
                }
#pragma unroll
            for (int i = 0; i < NDI; i++) {
                const int f = lane + i * kWave;
                if (f < C * 36) di[f] = qd[i];
            }
            if (lane < C * 6) rin[lane] = qi;
            __syncthreads();
            if (c + 1 < n_chunk) fetch(c + 1);
            const int csteps = (int)min((uint32_t)C, n - c * (uint32_t)C);
            for (int kk = 0; kk < csteps; kk++) {
                const long s = (long)c * C + kk;
                if (lane < 6) {
                    const double* t = tring + slot_k * 6;
                    const double* M = di + kk * 36;
                    double v;
                    if (fwd) v = M[lane * 6] * t[0] + M[lane * 6 + 1] * t[1] + M[lane * 6 + 2] * t[2] + M[lane * 6 + 3] * t[3] + M[lane * 6 + 4] * t[4] + M[lane * 6 + 5] * t[5];
                    else v = M[lane] * t[0] + M[6 + lane] * t[1] + M[12 + lane] * t[2] + M[18 + lane] * t[3] + M[24 + lane] * t[4] + M[30 + lane] * t[5];
                    yk[lane] = v;
                    xout[kk * 6 + lane] = v;
                }
                __builtin_amdgcn_s_barrier();   // one wave: a bare barrier orders the LDS traffic without draining the prefetch
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
                const double y0 = yk[0], y1 = yk[1], y2 = yk[2], y3 = yk[3], y4 = yk[4], y5 = yk[5];
                for (int e = lane; e < n_upd; e += kWave) {
                    const int d = 1 + e / 6, m = e % 6;
                    if (s + d < (long)n) {
                        int slot = slot_k + d;
                        if (slot >= W) slot -= W;
                        const double* l = rows + kk * rowlen + d * 36;
                        double acc;
                        if (fwd) {   // row m of L_{k+d,k}
                            const double* lr = l + m * 6;
                            acc = lr[0] * y0 + lr[1] * y1 + lr[2] * y2 + lr[3] * y3 + lr[4] * y4 + lr[5] * y5;
                        } else {     // column m of L_{k,k-d}
                            acc = l[m] * y0 + l[6 + m] * y1 + l[12 + m] * y2 + l[18 + m] * y3 + l[24 + m] * y4 + l[30 + m] * y5;
                        }
                        tring[slot * 6 + m] -= acc;
                    }
                }
                if (lane < 6) tring[slot_k * 6 + lane] = rin[kk * 6 + lane];   // the row of step s + W takes the freed slot
                slot_k = slot_k + 1 == W ? 0 : slot_k + 1;
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
            }
            for (int e = lane; e < csteps * 6; e += kWave) {
                const long k = row_of((long)c * C + e / 6);
                x[6 * (size_t)k + e % 6] = xout[e];
            }
        }
        __syncthreads();
    }
}

__host__ __device__ inline int solve_chunk(int bw) { return bw <= 10 ? 8 : 4; }
__host__ __device__ inline size_t solve_lds_count(int bw)
{
    const int C = solve_chunk(bw), W = bw + 1;
    return (size_t)C * W * 36 + (size_t)C * 36 + (size_t)C * 6 * 2 + (size_t)W * 6 + 8;
}

// PCG state shared between the round kernels (device f64 slots behind the work vectors)
enum { PS_RZ = 0, PS_BB = 1, PS_DONE = 2, PS_ROUNDS = 3, PS_PQ = 4, PS_COUNT = 8 };

// mode 0: plain solve x = M^-1 b.
// mode 1: PCG round head: z = M^-1 r (z = x argument), then p = z + beta p with beta from r.z (one wave, n small)
// mode 0: plain solve x = M^-1 b.
// mode 1: PCG round head: z = M^-1 r (z = x argument), then p = z + beta p with beta from r.z (one wave, n small)
__global__ __launch_bounds__(64) void band_solve_kernel(const uint32_t n, const int bw, const double* __restrict__ band,
                                                        const double* __restrict__ bandT, const double* __restrict__ dinv,
                                                        const double* __restrict__ b, double* __restrict__ x, const int mode,
                                                        double* __restrict__ p, double* __restrict__ state)
{
    extern __shared__ double lds[];
    if (mode == 1 && state[PS_DONE] != 0.0) return;
    if (bw <= 10) band_solve_wave<8, 7>(n, bw, band, bandT, dinv, b, x, lds);     // rows of <= 396 doubles
    else band_solve_wave<4, 9>(n, bw, band, bandT, dinv, b, x, lds);              // rows of <= 576 doubles
    if (mode == 1) {
        __syncthreads();
        const int lane = threadIdx.x;
        double rz = 0.0;
        for (uint32_t i = lane; i < n * 6; i += kWave) rz += b[i] * x[i];
        rz = wave_sum(rz);
        rz = __shfl(rz, 0, kWave);
        const double rz_old = state[PS_RZ];
        const bool first = state[PS_ROUNDS] == 0.0;   // p is uninitialised memory then: do not multiply it by zero
        const double beta = first ? 0.0 : rz / rz_old;
        for (uint32_t i = lane; i < n * 6; i += kWave) p[i] = first ? x[i] : x[i] + beta * p[i];
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

// PCG round head after an external preconditioner solve z = M^-1 r: r.z, beta, p = z + beta p (one workgroup)
__global__ __launch_bounds__(1024) void pcg_direction_kernel(uint32_t n, const double* __restrict__ r, const double* __restrict__ z,
                                                             double* __restrict__ p, double* __restrict__ state)
{
    __shared__ double red[16];
    if (state[PS_DONE] != 0.0) return;
    double rz = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) rz += r[i] * z[i];
    rz = block_sum(rz, red);
    const double beta = state[PS_ROUNDS] == 0.0 ? 0.0 : rz / state[PS_RZ];
    // first round: p is uninitialised memory, and 0 * garbage is not 0 when the garbage is a NaN
    const bool first = state[PS_ROUNDS] == 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) p[i] = first ? z[i] : z[i] + beta * p[i];
    __syncthreads();
    if (threadIdx.x == 0) state[PS_RZ] = rz;
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
// q = S p and the partial p.q: four lanes per scalar row share its block entries (the kernel is a latency-bound gather
// of 48-byte block rows), fixed butterfly, then the fixed block sum
constexpr int kMatvecLanes = 4;
__global__ __launch_bounds__(256) void pcg_band_matvec_kernel(const BsrView A, const double* __restrict__ p, double* __restrict__ q,
                                                              double* __restrict__ part, const double* __restrict__ state)
{
    __shared__ double red[4];
    if (state && state[PS_DONE] != 0.0) return;   // state == nullptr: first round of a solve, nothing to check yet
    const uint32_t n = A.n_rows * 6;
    const uint32_t i = (blockIdx.x * 256 + threadIdx.x) / kMatvecLanes, sub = threadIdx.x % kMatvecLanes;
    double s = 0.0;
    if (i < n) {
        const uint32_t f = i / 6, a = i % 6;
        for (uint32_t e = A.row_ptr[f] + sub; e < A.row_ptr[f + 1]; e += kMatvecLanes) {
            const double* B = A.blocks + 36 * (size_t)A.ent_blk[e];
            const double* x = p + 6 * (size_t)A.ent_col[e];
            if (A.ent_trans[e]) {
#pragma unroll
                for (int c = 0; c < 6; c++) s += B[c * 6 + a] * x[c];
            } else {
#pragma unroll
                for (int c = 0; c < 6; c++) s += B[a * 6 + c] * x[c];
            }
        }
    }
    static_assert(kMatvecLanes == 4, "quad_sum adds the four lanes of a row");
    s = quad_sum(s);
    double pq = 0.0;
    if (i < n && sub == 0) {
        q[i] = s;
        pq = p[i] * s;
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

// First PCG round of a solve with x0 = 0 (cyclic-reduction path): r = b, p = z, so there is no direction step, no
// initialisation pass and no copy of b - this kernel does what pcg_band_init, pcg_direction and pcg_band_update would:
// |b|^2, r.z, alpha = r.z / p.q, x = alpha z, r = b - alpha q, p = z, the state words, the convergence test.
__global__ __launch_bounds__(1024) void pcg_first_update_kernel(uint32_t n, uint32_t n_part, const double* __restrict__ part,
                                                                const double* __restrict__ b, const double* __restrict__ z,
                                                                const double* __restrict__ q, double* __restrict__ x,
                                                                double* __restrict__ r, double* __restrict__ p,
                                                                double* __restrict__ state, const double tol, double* __restrict__ scal)
{
    __shared__ double red[16];
    double pq = 0.0;
    for (uint32_t i = threadIdx.x; i < n_part; i += 1024) pq += part[i];
    pq = block_sum(pq, red);
    double bb = 0.0, rz = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const double bi = b[i];
        bb += bi * bi;
        rz += bi * z[i];
    }
    bb = block_sum(bb, red);
    rz = block_sum(rz, red);
    if (!(bb > 0.0) || !(pq > 0.0)) {
        // zero right-hand side: x = 0 is the answer; p.q <= 0 with b != 0: the matrix is not positive definite
        for (uint32_t i = threadIdx.x; i < n; i += 1024) { x[i] = 0.0; r[i] = b[i]; p[i] = z[i]; }
        if (threadIdx.x == 0) {
            state[PS_RZ] = rz; state[PS_BB] = bb; state[PS_DONE] = 1.0; state[PS_ROUNDS] = bb > 0.0 ? 1.0 : 0.0; state[PS_PQ] = 0.0;
            if (bb > 0.0) scal[SC_LIN_STATUS] = 2.0;
            scal[SC_LIN_ITERS] = state[PS_ROUNDS];
            scal[SC_LIN_RESID] = bb > 0.0 ? 1.0 : 0.0;
        }
        return;
    }
    const double alpha = rz / pq;
    double rr = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const double zi = z[i];
        x[i] = alpha * zi;
        const double ri = b[i] - alpha * q[i];
        r[i] = ri;
        p[i] = zi;
        rr += ri * ri;
    }
    rr = block_sum(rr, red);
    if (threadIdx.x == 0) {
        state[PS_RZ] = rz; state[PS_BB] = bb; state[PS_ROUNDS] = 1.0; state[PS_PQ] = 0.0;
        state[PS_DONE] = (rr > tol * tol * bb) ? 0.0 : 1.0;
        scal[SC_LIN_ITERS] = 1.0;
        scal[SC_LIN_RESID] = sqrt(rr / bb);
    }
}

}  // namespace

size_t band_count(uint32_t n_rows, int bw) { return (size_t)n_rows * (bw + 1) * 36; }
static uint32_t pcg_band_matvec_blocks(uint32_t n) { return (n * kMatvecLanes + 255) / 256; }
size_t pcg_band_work_count(uint32_t n_rows) { return (size_t)n_rows * 6 * 3 + PS_COUNT + pcg_band_matvec_blocks(n_rows * 6) + 8; }

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
    const int C = bw <= 9 ? 16 : 8;   // pivot rows per LDS chunk: 75 KB at bw = 9, 110 KB at bw = 15
    const size_t lds = sizeof(double) * ((size_t)(bw + 1 + C) * (bw + 1) * 36 + (size_t)C * 72);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(band_cholesky_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(band_cholesky_kernel, dim3(1), dim3(kFactorThreads), lds, s, band, bandT, n_rows, bw, C, dinv, scal);
}

void launch_band_solve(hipStream_t s, uint32_t n_rows, int bw, const double* band, const double* bandT, const double* dinv,
                       const double* b, double* x)
{
    if (!n_rows) return;
    const size_t lds = sizeof(double) * solve_lds_count(bw);
    hipLaunchKernelGGL(band_solve_kernel, dim3(1), dim3(64), lds, s, n_rows, bw, band, bandT, dinv, b, x, 0, nullptr, nullptr);
}

void launch_pcg_band(hipStream_t s, const BsrView& A, int bw, const double* band, const double* bandT, const double* dinv,
                     const double* b, double* x, double* resid, double* work, double tol, int max_rounds, double* scal)
{
    if (!A.n_rows) return;
    const uint32_t n = A.n_rows * 6, n_wg = pcg_band_matvec_blocks(n);
    double* p = work;
    double* z = work + n;
    double* q = work + 2 * (size_t)n;
    double* state = work + 3 * (size_t)n;
    double* part = state + PS_COUNT;
    hipLaunchKernelGGL(pcg_band_init_kernel, dim3(1), dim3(1024), 0, s, n, b, x, resid, state);
    for (int round = 0; round < max_rounds; round++) {
        // every kernel of a round returns at once when the previous round converged (device-side flag)
        hipLaunchKernelGGL(band_solve_kernel, dim3(1), dim3(64), sizeof(double) * solve_lds_count(bw), s, A.n_rows, bw, band, bandT, dinv,
                           resid, z, 1, p, state);
        hipLaunchKernelGGL(pcg_band_matvec_kernel, dim3(n_wg), dim3(256), 0, s, A, p, q, part, state);
        hipLaunchKernelGGL(pcg_band_update_kernel, dim3(1), dim3(1024), 0, s, n, n_wg, part, p, q, x, resid, state, tol, scal);
    }
}

void launch_pcg_cr(hipStream_t s, const BsrView& A, int bw, double* cr_ws, const double* b, double* x, double* resid, double* work,
                   double tol, int max_rounds, double* scal, bool forward_done)
{
    if (!A.n_rows || max_rounds < 1) return;
    const uint32_t n = A.n_rows * 6, n_wg = pcg_band_matvec_blocks(n);
    double* p = work;
    double* z = work + n;
    double* q = work + 2 * (size_t)n;
    double* state = work + 3 * (size_t)n;
    double* part = state + PS_COUNT;
    // round 0 (x0 = 0, r = b, p = z): three steps, none of which looks at the state of the previous solve
    launch_cr_solve(s, A.n_rows, bw, cr_ws, b, z, nullptr, forward_done);   // the factorisation may have carried b down the tree
    hipLaunchKernelGGL(pcg_band_matvec_kernel, dim3(n_wg), dim3(256), 0, s, A, z, q, part, nullptr);
    hipLaunchKernelGGL(pcg_first_update_kernel, dim3(1), dim3(1024), 0, s, n, n_wg, part, b, z, q, x, resid, p, state, tol, scal);
    launch_pcg_cr_more(s, A, bw, cr_ws, x, resid, work, tol, max_rounds - 1, scal);
}

// further rounds of a solve launch_pcg_cr began (its state lies in work): each returns at once when an earlier round converged
void launch_pcg_cr_more(hipStream_t s, const BsrView& A, int bw, double* cr_ws, double* x, double* resid, double* work, double tol, int rounds,
                        double* scal)
{
    if (!A.n_rows) return;
    const uint32_t n = A.n_rows * 6, n_wg = pcg_band_matvec_blocks(n);
    double* p = work;
    double* z = work + n;
    double* q = work + 2 * (size_t)n;
    double* state = work + 3 * (size_t)n;
    double* part = state + PS_COUNT;
    for (int round = 0; round < rounds; round++) {
        launch_cr_solve(s, A.n_rows, bw, cr_ws, resid, z, state + PS_DONE);
        hipLaunchKernelGGL(pcg_direction_kernel, dim3(1), dim3(1024), 0, s, n, resid, z, p, state);
        hipLaunchKernelGGL(pcg_band_matvec_kernel, dim3(n_wg), dim3(256), 0, s, A, p, q, part, state);
        hipLaunchKernelGGL(pcg_band_update_kernel, dim3(1), dim3(1024), 0, s, n, n_wg, part, p, q, x, resid, state, tol, scal);
    }
}

}  // namespace soslam
