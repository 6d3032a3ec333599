// Development probe (GPU box): cycles and wall time of the register-resident Gauss-Jordan sweep of crsolve.hip.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I stereo_orb_slam_amd/csrc -mllvm -amdgpu-mfma-vgpr-form \
//        scripts/probes/gj_probe.hip -o gpurun_out/gj_probe
#include "../../stereo_orb_slam_amd/csrc/crsolve.hip"

#include <cstdio>
#include <vector>

using namespace soslam;
namespace soslam { namespace {
__device__ unsigned long long g_stamps[8];
#define MFMA_X(a_, b_, c_, x0, x1, x2) ((MODE & 1) ? (c_) + (a_) * (b_) * 0.0 : __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, c_, 0, 0, 0))
template <int NSTEPS, int MODE>
__device__ __forceinline__ bool gj_sweep_x(double4_t (&acc)[4], double (*strips)[4][64], double (*pblk)[16], const int wave, const int lane)
{
    const int lr = lane % 16, lk = lane / 16;
    const int g = lk < 3 ? lk : 0;                       // the padding group computes group 0's row and multiplies zeros with it
    const int p0 = g, p1 = (g + 1) % 3, p2 = (g + 2) % 3;
    // the pivot block is read from its upper triangle only (the accumulators are symmetric up to rounding; every group must
    // see the same numbers)
    const int u01 = (p0 < p1 ? p0 : p1) * 3 + (p0 < p1 ? p1 : p0), u02 = (p0 < p2 ? p0 : p2) * 3 + (p0 < p2 ? p2 : p0),
              u12 = (p1 < p2 ? p1 : p2) * 3 + (p1 < p2 ? p2 : p1);
    // strip rows this lane reads: its B operand row (zeros for the padding k = 3) and the three pivot rows in permuted order
    const int zb = (lk < 3 ? lk : 3) * 64 + 16 * wave + lr;
    const int za0 = (lk < 3 ? p0 : 3) * 64 + lr, za1 = (lk < 3 ? p1 : 3) * 64 + lr, za2 = (lk < 3 ? p2 : 3) * 64 + lr;
    const int jcol = 16 * wave + lr;                     // the column this lane holds
    if (threadIdx.x < 128) strips[threadIdx.x / 64][3][threadIdx.x % 64] = 0.0;
    bool ok = true;
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
                strip[c * 64 + jcol] = val;
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
                strip[c * 64 + jcol] = val;
                acc[qb / 4][qb % 4] = 0.0;
            }
        }
    };
    publish(0);
    double aop_t[4] = {0.0, 0.0, 0.0, 0.0}, sv_t = 0.0;   // operands of the updates the previous step left for this one
#pragma unroll
    for (int kb = 0; kb < NSTEPS; kb++) {   // straight-line code: every register index below is static
        const int k0 = 3 * kb;
        const double* strip = &strips[kb & 1][0][0];
        const double* pb = &pblk[kb & 1][0];
        if (!(MODE & 8)) __syncthreads();
        // requests first, arithmetic behind them
        const double a = pb[p0 * 4], b = pb[u01], c = pb[u02], d = pb[p1 * 4], e = pb[u12], f = pb[p2 * 4];
        double ac[4][3];
#pragma unroll
        for (int tr = 0; tr < 4; tr++) { if (MODE & 16) { ac[tr][0] = a; ac[tr][1] = b; ac[tr][2] = d; } else { ac[tr][0] = strip[za0 + 16 * tr]; ac[tr][1] = strip[za1 + 16 * tr]; ac[tr][2] = strip[za2 + 16 * tr]; } }
        const double sv = strip[zb];
        // the previous step's updates of the tiles that did not hold this step's pivot rows: their matrix instructions run
        // while the requests above are under way
        if (kb > 0) {
            const int m0 = k0 / 16, m1 = (k0 + 2) / 16;   // the tile rows the previous step updated before its exchange
#pragma unroll
            for (int tr = 0; tr < 4; tr++)
                if (tr != m0 && tr != m1) acc[tr] = MFMA_X(aop_t[tr], sv_t, acc[tr], 0, 0, 0);
        }
        // row 0 of the inverse of the permuted block = row lk of Pinv in the column order p0 p1 p2
        const double c00 = __builtin_fma(d, f, -(e * e)), c01 = __builtin_fma(c, e, -(b * f)), c02 = __builtin_fma(b, e, -(c * d));
        const double det_own = __builtin_fma(a, c00, __builtin_fma(b, c01, c * c02));
        const double det = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(det_own)),
                                            __builtin_amdgcn_readfirstlane(__double2loint(det_own)));
        ok = ok && (a > 0.0) && (c00 > 0.0) && (det > 0.0);
        double id = __builtin_amdgcn_rcp(det);
        id = id * (2.0 - det * id);
        id = id * (2.0 - det * id);
        double r0 = c00 * id, r1 = c01 * id, r2 = c02 * id;
        if (MODE & 2) { r0 = a; r1 = b; r2 = c; }
        const int cj = jcol - k0;                        // 0..2: this lane holds a pivot column: base zero
        const bool pcol = cj >= 0 && cj < 3;
        // A'[i][k = lk], i = 16 tr + lr
        auto operand = [&](const int tr) __attribute__((always_inline)) {
            const double v = ac[tr][0] * r0 + ac[tr][1] * r1 + ac[tr][2] * r2;
            return 16 * tr + lr < k0 + 3 ? v : -v;
        };
        const bool more = kb + 1 < NSTEPS;
        const int n0 = more ? (k0 + 3) / 16 : -1, n1 = more ? (k0 + 5) / 16 : -1;   // static: the tile rows of the next pivot rows
        // pivot columns start from zero: only the wave(s) that hold them (a scalar branch), the lanes under the execution mask
        if (wave == k0 / 16 || wave == (k0 + 2) / 16) {
            if (pcol) {
#pragma unroll
                for (int tr = 0; tr < 4; tr++)
#pragma unroll
                    for (int r = 0; r < 4; r++) asm volatile("v_mov_b64 %0, 0" : "=v"(acc[tr][r]));
            }
        }
        if (more) {
            // the tile row(s) of the next pivot rows now, their exchange behind them; the other tiles' operands are kept
            acc[n0] = MFMA_X(operand(n0), sv, acc[n0], 0, 0, 0);
            if (n1 != n0) acc[n1] = MFMA_X(operand(n1), sv, acc[n1], 0, 0, 0);
#pragma unroll
            for (int tr = 0; tr < 4; tr++)
                if (tr != n0 && tr != n1) aop_t[tr] = operand(tr);
            sv_t = sv;
            __builtin_amdgcn_sched_barrier(0);   // the operands above fill the wait for the matrix instruction's result
            if (!(MODE & 4)) publish(kb + 1);
        } else {
#pragma unroll
            for (int tr = 0; tr < 4; tr++) acc[tr] = MFMA_X(operand(tr), sv, acc[tr], 0, 0, 0);
        }
    }
    return ok;
}



} }

template <int NSTEPS, int MODE>
__global__ __launch_bounds__(256) void probe_kernel(const double* __restrict__ D, double* __restrict__ E, unsigned long long* __restrict__ t, int reps)
{
    __shared__ double strips[2][4][64];
    __shared__ double pblk[2][16];
    const int lane = threadIdx.x % 64, wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    const int lr = lane % 16, lk = lane / 16;
    double4_t acc[4];
    unsigned long long c_sum = 0, r_sum = 0;
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[tj][r] = D[(size_t)blockIdx.x * 4096 + (16 * tj + 4 * r + lk) * 64 + 16 * wave + lr];   // wave w: tile column w
        __syncthreads();
        const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
        const bool ok = MODE < 0 ? gj_sweep<NSTEPS>(acc, strips, pblk, wave, lane) : gj_sweep_x<NSTEPS, (MODE < 0 ? 0 : MODE)>(acc, strips, pblk, wave, lane);
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        c_sum += c1 - c0; r_sum += r1 - r0;
        if (!ok) E[0] = -1.0;
        __syncthreads();
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int r = 0; r < 4; r++) E[(size_t)blockIdx.x * 4096 + (16 * tj + 4 * r + lk) * 64 + 16 * wave + lr] = acc[tj][r];
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c_sum; t[2 * blockIdx.x + 1] = r_sum; }
}


int main()
{
    const int nb = 28, reps = 50;
    std::vector<double> h((size_t)nb * 4096, 0.0);
    // SPD: A = G G^T + 64 I from a fixed pseudo-random G
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s % 2000001) / 1e6 - 1.0; };
    for (int b = 0; b < nb; b++) {
        std::vector<double> G(64 * 64);
        for (auto& g : G) g = rnd();
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 64; j++) {
                double v = i == j ? 64.0 : 0.0;
                for (int k = 0; k < 64; k++) v += G[i * 64 + k] * G[j * 64 + k];
                h[(size_t)b * 4096 + i * 64 + j] = v;
            }
    }
    double *dD, *dE; unsigned long long* dt;
    hipMalloc(&dD, h.size() * 8); hipMalloc(&dE, h.size() * 8); hipMalloc(&dt, nb * 16);
    hipMemcpy(dD, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int mode : {-1, 0}) {
        const int grid = 1;
        for (int it = 0; it < 2; it++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (mode) {
            case -1: hipLaunchKernelGGL((probe_kernel<20, -1>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 0: hipLaunchKernelGGL((probe_kernel<20, 0>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 1: hipLaunchKernelGGL((probe_kernel<20, 1>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 2: hipLaunchKernelGGL((probe_kernel<20, 2>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 3: hipLaunchKernelGGL((probe_kernel<20, 3>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 4: hipLaunchKernelGGL((probe_kernel<20, 4>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 5: hipLaunchKernelGGL((probe_kernel<20, 5>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 7: hipLaunchKernelGGL((probe_kernel<20, 7>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 8: hipLaunchKernelGGL((probe_kernel<20, 8>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 16: hipLaunchKernelGGL((probe_kernel<20, 16>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            case 23: hipLaunchKernelGGL((probe_kernel<20, 23>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            default: hipLaunchKernelGGL((probe_kernel<20, 31>), dim3(grid), dim3(256), 0, 0, dD, dE, dt, reps); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t[2]; hipMemcpy(t, dt, 16, hipMemcpyDeviceToHost);
            printf("mode %2d grid %2d: kernel %.1f us for %d sweeps of 20 steps; per sweep %.0f cycles, %.2f us (memrealtime) -> %.0f cycles/step, clock %.2f GHz\n",
                   mode, grid, ms * 1e3, reps, (double)t[0] / reps, (double)t[1] / reps * 0.01, (double)t[0] / reps / 20, (double)t[0] / ((double)t[1] * 10.0));
        }
    }
    { unsigned long long st[8]; hipMemcpyFromSymbol(st, HIP_SYMBOL(soslam::g_stamps), sizeof st); const double n = 2.0 * reps * 17;   // steps 2..18 of two launches
      printf("stamps (cycles per step, wave 0): to-barrier %.0f  barrier %.0f  reads %.0f  inverse %.0f  first-mfma %.0f  publish %.0f  trailing %.0f\n", st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, st[5] / n, st[6] / n); }
    // check: E D = I for block 0 (one sweep applied 'reps' times alternates A -> A^-1 -> A ...; reps even => E == D)
    std::vector<double> e(4096);
    hipLaunchKernelGGL((probe_kernel<20, 0>), dim3(1), dim3(256), 0, 0, dD, dE, dt, 1);
    hipMemcpy(e.data(), dE, 4096 * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 60; i++)
        for (int j = 0; j < 60; j++) {
            double v = 0;
            for (int k = 0; k < 60; k++) v += e[i * 64 + k] * h[k * 64 + j];
            worst = fmax(worst, fabs(v - (i == j ? 1.0 : 0.0)));
        }
    printf("max |E D - I| over the leading 60 x 60 block = %.3e (the sweep of 20 steps inverts it; rows 60..63 stay)\n", worst);
    return 0;
}
