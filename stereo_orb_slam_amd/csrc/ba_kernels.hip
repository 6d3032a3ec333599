// ba_kernels.hip - HIP kernels of the bundle-adjustment hot path for gfx950 (CDNA4, wave64).
//
// Data layout in HBM:
//   uv      [n_obs] float4          camera-major observation order (bucketed by camera)
//   obs_pt  [n_obs] u32             internal point id (points sorted by first camera)
//   ar      [n_obs][10] f64         compact row [G = A^T A (6) | h = A^T r (3) | 0], 80-B rows, camera-major like uv;
//                                   the products of J_p = A R and J_c = [A D | A] are rebuilt from it by the consumers
//                                   (ba_device.h); point-major passes gather rows through the pt_obs index
//   C/gp    [n_pt][6] / [n_pt][3]   per-point J_p^T J_p (xx xy xz yy yz zz) and J_p^T r
//   S       [n_blocks][36]          upper block-sparse reduced camera matrix (6x6 row-major blocks)
// The kernels are HBM-bound streaming / gather work at a few flop/B; the one GEMM-shaped piece, the Schur window
// product, runs on the f64 matrix cores.
#include "ba_kernels.h"
#include "reduce.h"

namespace soslam {

namespace {

constexpr int kWave = 64;

// The sums of N <= 32 values over the wave by a halving butterfly: at step k a lane keeps the half of its list that its bit
// k selects and adds the partner's (lane ^ 2^k) copy of that half - 16 + 8 + 4 + 2 + 1 exchanges and one last add instead
// of 6 N.  Every lane returns the sum of value number `index` (the low five lane bits reversed); fixed order of additions.
template <int N, int K>
__device__ __forceinline__ void wave_sum_table_step(double (&w)[32], const int lane)
{
    constexpr int half = 16 >> K;
    const bool up = ((lane >> K) & 1) != 0;
#pragma unroll
    for (int i = 0; i < half; i++) {
        const double a = w[i], b = w[half + i];
        const double send = up ? a : b, keep = up ? b : a;
        w[i] = keep + lane_xor_pow2<K>(send);
    }
}

template <int N>
__device__ __forceinline__ double wave_sum_table(const double (&v)[N], const int lane, int& index)
{
    static_assert(N <= 32, "one value per lane of a half wave");
    double w[32];
#pragma unroll
    for (int i = 0; i < 32; i++) w[i] = i < N ? v[i] : 0.0;
    wave_sum_table_step<N, 0>(w, lane);
    wave_sum_table_step<N, 1>(w, lane);
    wave_sum_table_step<N, 2>(w, lane);
    wave_sum_table_step<N, 3>(w, lane);
    wave_sum_table_step<N, 4>(w, lane);
    const F64Pair p = swap_halves32(w[0]);
    index = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
    return p.a + p.b;
}

// per-camera rotation block (R, R*Jr, t, branch flag) - 24 f64 per camera, recomputed whenever poses change
__global__ __launch_bounds__(64) void ba_pose_prepare_kernel(uint32_t n_cam, const double* __restrict__ cams,
                                                             double* __restrict__ campre)
{
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c < n_cam) pose_prepare(cams + 6 * (size_t)c, campre + kPoseStride * (size_t)c);
}

// ---------------------------------------------------------------------------------------------------
// K1  ba_linearize: one workgroup (256 lanes) per camera tile of <= 1024 observations of ONE camera, so the
// camera's rotation block is workgroup-uniform and arrives through scalar loads.  Each lane evaluates up
// to 4 observations (strided by 256, coalesced), writes their corrected Jacobian rows, and keeps its share
// of J_c^T J_c / J_c^T r / rho in registers; one wave-shuffle + LDS reduction per workgroup.
// Only the compact row [G | h] (ba_device.h) leaves the kernel: 16 (uv) + 4 (point id) + 24 (point) in, 80 out
// = 124 B per observation ("fused" in the sense of SURVEY.md section 8(d): J_c and J_p are never written).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTileThreads) void ba_linearize_kernel(
    const Tile* __restrict__ tiles, const float4* __restrict__ uv, const uint32_t* __restrict__ obs_pt,
    const double* __restrict__ campre, const double* __restrict__ pts, const int32_t* __restrict__ cam_free,
    const Proj P, const double delta, const CompactRows ar_out, double* __restrict__ tile_part,
    const double* __restrict__ gate)
{
    if (gate && *gate == 0.0) return;   // speculative launch (see ba_step_sums_kernel): the step was not accepted
    // Row-per-lane stores would touch 64 different lines per instruction; instead each wave stages its 64 rows
    // in LDS (the G rows, then the h rows) and writes them back out as whole 1-KiB pieces, 16 contiguous bytes per lane.
    constexpr int kArPad = kArG + kArH;   // staging doubles per row
    __shared__ double red[(kTileThreads / kWave) * kTileVals];
    __shared__ __attribute__((aligned(16))) double stage[(kTileThreads / kWave) * kWave * kArPad];
    const Tile t = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const int wave = tid / kWave, lane = tid % kWave;
    const bool fixed = cam_free[t.cam] < 0;
    PosePre pr;
    pose_load(campre + kPoseStride * (size_t)t.cam, pr);
    double* wstage = stage + wave * kWave * kArPad;

    double v[kTileVals];
#pragma unroll
    for (int i = 0; i < kTileVals; i++) v[i] = 0.0;

    // The lane's (up to four) observations are a chain of dependent gathers (index -> point) in front of a long
    // arithmetic block: all measurements and indices are requested up front, and each point one observation ahead.
    constexpr int kObsPerLane = kTileObs / kTileThreads;
    float4 m_all[kObsPerLane];
    uint32_t p_all[kObsPerLane];
#pragma unroll
    for (int j = 0; j < kObsPerLane; j++) {
        const uint32_t o = (uint32_t)j * kTileThreads + tid;
        const size_t k = (size_t)t.start + (o < t.count ? o : 0);
        m_all[j] = uv[k];
        p_all[j] = obs_pt[k];
    }
    double xn[3] = {pts[3 * (size_t)p_all[0]], pts[3 * (size_t)p_all[0] + 1], pts[3 * (size_t)p_all[0] + 2]};

#pragma unroll
    for (int j = 0; j < kObsPerLane; j++) {
        const uint32_t base = (uint32_t)j * kTileThreads;
        if (base >= t.count) break;   // uniform
        const uint32_t o = base + tid;
        const bool act = o < t.count;
        const uint32_t wave_first = base + wave * kWave;                     // first observation of this wave's 64 rows
        const int n_rows = wave_first < t.count ? (int)min(t.count - wave_first, (uint32_t)kWave) : 0;
        double r[4], am[12], G[6] = {0, 0, 0, 0, 0, 0}, hh[3] = {0, 0, 0};
        const double x[3] = {xn[0], xn[1], xn[2]};
        if (j + 1 < kObsPerLane) {
            const size_t pn = p_all[j + 1 < kObsPerLane ? j + 1 : j];
            xn[0] = pts[3 * pn]; xn[1] = pts[3 * pn + 1]; xn[2] = pts[3 * pn + 2];
        }
        if (act) {
            double vv[3];
            v[27] += residual_av(pr, x, m_all[j], P, delta, r, am, vv);
            // G = A^T A, h = A^T r: the compact row, and all the camera sums need
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double a0 = am[i * 3], a1 = am[i * 3 + 1], a2 = am[i * 3 + 2];
                G[0] += a0 * a0; G[1] += a0 * a1; G[2] += a0 * a2; G[3] += a1 * a1; G[4] += a1 * a2; G[5] += a2 * a2;
                hh[0] += a0 * r[i]; hh[1] += a1 * r[i]; hh[2] += a2 * r[i];
            }
            if (!fixed) {
                // J_c = A [D | I], D = -E M with E = [vv]x and M the camera's constant:  D^T G D = M^T (E G E^T) M,
                // D^T G = M^T (E G), D^T h = M^T (E h).  The tile sums hold the parts WITHOUT M - E G E^T, E G, G, E h, h,
                // in the places of the 6x6 upper triangle and the gradient - and ba_schur_reduce applies M once per camera.
                const double Gf[9] = {G[0], G[1], G[2], G[1], G[3], G[4], G[2], G[4], G[5]};
                double Z[9];   // E G: column c = vv x G[:, c]
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    Z[0 + c] = vv[1] * Gf[6 + c] - vv[2] * Gf[3 + c];
                    Z[3 + c] = vv[2] * Gf[0 + c] - vv[0] * Gf[6 + c];
                    Z[6 + c] = vv[0] * Gf[3 + c] - vv[1] * Gf[0 + c];
                }
                // E G E^T = -Z E: row a = vv x Z[a, :]
                double Q[9];
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    Q[a * 3 + 0] = vv[1] * Z[a * 3 + 2] - vv[2] * Z[a * 3 + 1];
                    Q[a * 3 + 1] = vv[2] * Z[a * 3 + 0] - vv[0] * Z[a * 3 + 2];
                    Q[a * 3 + 2] = vv[0] * Z[a * 3 + 1] - vv[1] * Z[a * 3 + 0];
                }
                // upper triangle in the order (0,0) (0,1) .. (0,5) (1,1) .. (5,5)
                int idx = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++) {
                        double val;
                        if (a < 3 && b < 3) val = Q[a * 3 + b];
                        else if (a < 3) val = Z[a * 3 + (b - 3)];
                        else val = Gf[(a - 3) * 3 + (b - 3)];
                        v[idx++] += val;
                    }
                v[21] += vv[1] * hh[2] - vv[2] * hh[1];
                v[22] += vv[2] * hh[0] - vv[0] * hh[2];
                v[23] += vv[0] * hh[1] - vv[1] * hh[0];
#pragma unroll
                for (int a = 0; a < 3; a++) v[24 + a] += hh[a];
            }
        }
        if (n_rows > 0) {
            const size_t row0 = (size_t)t.start + wave_first;
            // G = A^T A (64 rows x 6 doubles, 3 KiB) and [h | 0], h = A^T r (64 rows x 4 doubles, 2 KiB): two contiguous blocks
            if (act) {
                double2* wg = reinterpret_cast<double2*>(wstage + lane * kArG);
                double2* wh = reinterpret_cast<double2*>(wstage + kWave * kArG + lane * kArH);
                wg[0] = make_double2(G[0], G[1]); wg[1] = make_double2(G[2], G[3]); wg[2] = make_double2(G[4], G[5]);
                wh[0] = make_double2(hh[0], hh[1]); wh[1] = make_double2(hh[2], 0.0);
            }
            __builtin_amdgcn_wave_barrier();
            {
                double2* g = reinterpret_cast<double2*>(ar_out.g + kArG * row0);
                double2* hrow = reinterpret_cast<double2*>(ar_out.h + kArH * row0);
                const double2* sg = reinterpret_cast<const double2*>(wstage);
                const double2* sh = reinterpret_cast<const double2*>(wstage + kWave * kArG);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int c = i * kWave + lane;           // 16-byte piece of the wave's 3-KiB block of G rows
                    if (c / 3 < n_rows) g[c] = sg[c];
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int c = i * kWave + lane;           // 16-byte piece of the wave's 2-KiB block of h rows
                    if (c / 2 < n_rows) hrow[c] = sh[c];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    {
        int idx;
        const double sum = wave_sum_table(v, lane, idx);
        if (lane < 32 && idx < kTileVals) red[wave * kTileVals + idx] = sum;
    }
    __syncthreads();
    if (tid < kTileVals) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kTileThreads / kWave; w++) sum += red[w * kTileVals + tid];
        tile_part[(size_t)blockIdx.x * kTileVals + tid] = sum;
    }
}

// Read-back for the parity tests only: the loss-corrected residual and the full Jacobian blocks of every observation,
// formed from the SAME function ba_linearize calls (residual_ad: r, A, D) - J_p = A R, J_c = [A D | A] - so the blocks the
// tests compare are those the shipped kernel folds into the compact row.  (The stored rows themselves are read back
// through SOSLAM_DBG_COMPACT_ROWS.)
__global__ __launch_bounds__(kTileThreads) void ba_debug_rows_kernel(
    const Tile* __restrict__ tiles, const float4* __restrict__ uv, const uint32_t* __restrict__ obs_pt,
    const double* __restrict__ campre, const double* __restrict__ pts, const int32_t* __restrict__ cam_free,
    const Proj P, const double delta, double* __restrict__ r_out, double* __restrict__ jc_out, double* __restrict__ jp_out)
{
    const Tile t = tiles[blockIdx.x];
    const bool fixed = cam_free[t.cam] < 0;
    PosePre pr;
    pose_load(campre + kPoseStride * (size_t)t.cam, pr);
    for (uint32_t o = threadIdx.x; o < t.count; o += kTileThreads) {
        const size_t k = (size_t)t.start + o;
        const uint32_t p = obs_pt[k];
        const double x[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
        double r[4], am[12], D[9];
        (void)residual_ad(pr, x, uv[k], P, delta, r, am, D);
        for (int i = 0; i < 4; i++) {
            r_out[4 * k + i] = r[i];
            const double a0 = am[i * 3], a1 = am[i * 3 + 1], a2 = am[i * 3 + 2];
            for (int j = 0; j < 3; j++) {
                jc_out[24 * k + i * 6 + j] = fixed ? 0.0 : a0 * D[j] + a1 * D[3 + j] + a2 * D[6 + j];
                jc_out[24 * k + i * 6 + 3 + j] = fixed ? 0.0 : am[i * 3 + j];
                jp_out[12 * k + i * 3 + j] = a0 * pr.R[j] + a1 * pr.R[3 + j] + a2 * pr.R[6 + j];
            }
        }
    }
}

// K2  ba_cost: residual + loss only, same tiling.  48 B per observation in, 8 B per tile out.
__global__ __launch_bounds__(kTileThreads) void ba_cost_kernel(
    const Tile* __restrict__ tiles, const float4* __restrict__ uv, const uint32_t* __restrict__ obs_pt,
    const double* __restrict__ campre, const double* __restrict__ pts, const Proj P, const double delta,
    double* __restrict__ cost_part)
{
    __shared__ double red[kTileThreads / kWave];
    const Tile t = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    PosePre pr;
    pose_load(campre + kPoseStride * (size_t)t.cam, pr);
    double rho = 0.0;
    for (uint32_t o = tid; o < t.count; o += kTileThreads) {
        const size_t k = (size_t)t.start + o;
        const uint32_t p = obs_pt[k];
        const double x[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
        rho += residual_cost(pr, x, uv[k], P, delta);
    }
    const double sum = wave_sum(rho);
    if (tid % kWave == 0) red[tid / kWave] = sum;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kTileThreads / kWave; w++) s += red[w];
        cost_part[blockIdx.x] = s;
    }
}

// fixed-order sum of a strided column: one workgroup, each lane a fixed subsequence, then a fixed tree
struct StepGate {   // the acceptance test of the LM loop for ba_step_sums_kernel (gate == nullptr: none)
    double* gate;            // out: 1.0 accepted, 0.0 not (or speculation disabled)
    const double* status;    // scal + SC_LIN_ITERS: [2] = SC_LIN_STATUS, [3] = SC_SCHUR_STATUS
    double x_cost, min_relative_decrease;
    int enabled;
    int armijo;              // also require the sufficient-decrease condition of the bounded-problem line search (the host
                             // shortens the step when it fails, so the candidate linearised behind the gate would be the wrong one)
};

__global__ __launch_bounds__(1024) void sum_strided_kernel(const double* __restrict__ in, uint32_t n, uint32_t stride,
                                                            uint32_t offset, double scale, double* __restrict__ out, const Publish pb)
{
    __shared__ double red[16];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) s += in[(size_t)i * stride + offset];
    s = wave_sum(s);
    if (threadIdx.x % kWave == 0) red[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; w++) t += red[w];
        out[0] = scale * t;
    }
    publish_tail(pb);
}

__global__ __launch_bounds__(1024) void sum5_kernel(const double* __restrict__ in, uint32_t n, double* __restrict__ out)
{
    __shared__ double red[16 * 5];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, mx = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const double* v = in + 5 * (size_t)i;
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        mx = fmax(mx, v[4]);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); mx = wave_max(mx);
    if (threadIdx.x % kWave == 0) {
        double* o = red + (threadIdx.x / kWave) * 5;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0, d = 0.0, e = 0.0;
        for (int w = 0; w < 16; w++) { a += red[w * 5]; b += red[w * 5 + 1]; c += red[w * 5 + 2]; d += red[w * 5 + 3]; e = fmax(e, red[w * 5 + 4]); }
        out[0] = a; out[1] = b; out[2] = c; out[3] = d; out[4] = e;
    }
}

// The two one-workgroup sums that end an iteration in one launch: the back-substitution's five step scalars (as
// sum5_kernel) and the candidate cost (as sum_strided_kernel, stride 1, scale 0.5), then the optional publication.
// The sums behind a step - the back-substitution's and the camera update's step scalars, the candidate cost -, the acceptance test
// and the publication, for NT lanes.  SC1: the partials were written by other workgroups of the SAME launch (ba_apply_small's
// last workgroup runs this): every load of them is an sc1 load.
struct StepSumsArgs {
    const double* part5; uint32_t n5; double* out5;
    const double* cam5; uint32_t n_cam5; double* out_cam5;
    const double* cost_part; uint32_t n_cost; double* out_cost;
    StepGate sg; double stop_vote; Publish pb;
};

template <int NT, bool SC1>
__device__ __forceinline__ void step_sums_body(const StepSumsArgs& q, double* red /* [NT / 64][11] */, double* fin /* [11] */)
{
    auto ld = [](const double* ptr) __attribute__((always_inline)) -> double {
        if constexpr (SC1) return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *ptr;
    };
    constexpr int NW = NT / kWave;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, mx = 0.0, sc = 0.0;
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0, cx = 0.0;
    for (uint32_t i = threadIdx.x; i < q.n5; i += NT) {
        const double* v = q.part5 + 5 * (size_t)i;
        s0 += ld(v); s1 += ld(v + 1); s2 += ld(v + 2); s3 += ld(v + 3);
        mx = fmax(mx, ld(v + 4));
    }
    for (uint32_t i = threadIdx.x; i < q.n_cam5; i += NT) {
        const double* v = q.cam5 + 5 * (size_t)i;
        c0 += ld(v); c1 += ld(v + 1); c2 += ld(v + 2); c3 += ld(v + 3);
        cx = fmax(cx, ld(v + 4));
    }
    for (uint32_t i = threadIdx.x; i < q.n_cost; i += NT) sc += ld(q.cost_part + i);
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); mx = wave_max(mx); sc = wave_sum(sc);
    c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2); c3 = wave_sum(c3); cx = wave_max(cx);
    if (threadIdx.x % kWave == 0) {
        double* o = red + (threadIdx.x / kWave) * 11;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = mx; o[5] = sc;
        o[6] = c0; o[7] = c1; o[8] = c2; o[9] = c3; o[10] = cx;
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        // lane k adds (or maximises) value k of the wave partials, in wave order
        const int k = threadIdx.x;
        const bool is_max = k == 4 || k == 10;
        double a = 0.0;
        for (int w = 0; w < NW; w++) a = is_max ? fmax(a, red[w * 11 + k]) : a + red[w * 11 + k];
        if (k == 5) a *= 0.5;
        fin[k] = a;
        if (k < 5) q.out5[k] = a;
        else if (k == 5) q.out_cost[0] = a;
        else q.out_cam5[k - 6] = a;
        if (k == 0) q.out5[5] = q.stop_vote;   // SC_STOP
    }
    if (q.sg.gate) {
        // The LM loop's acceptance test, on the device: the host has already enqueued the linearisation at the candidate
        // behind this kernel, gated by this word, so an accepted step goes on without waiting for the host (which reads
        // the same word and follows it).  Same operands and operations as run_lm.
        __syncthreads();
        if (threadIdx.x == 0) {
            const double mcc = fin[0] + fin[6], cand = fin[5], gdot = fin[3] + fin[9];
            const bool ok = q.sg.status[2] == 0.0 && q.sg.status[3] == 0.0 && isfinite(mcc) && isfinite(cand) && mcc > 0.0 &&
                            (!q.sg.armijo || cand <= q.sg.x_cost + 1e-4 * gdot);
            *q.sg.gate = (q.sg.enabled && ok && (q.sg.x_cost - cand) / mcc > q.sg.min_relative_decrease) ? 1.0 : 0.0;
        }
    }
    publish_tail(q.pb);
}

__global__ __launch_bounds__(1024) void ba_step_sums_kernel(const StepSumsArgs q)
{
    __shared__ double red[16 * 11];
    __shared__ double fin[11];
    step_sums_body<1024, false>(q, red, fin);
}

// The acceptance test on scalars that were summed over the ranks after ba_step_sums wrote them, then the publication.
__global__ __launch_bounds__(64) void ba_gate_publish_kernel(double* __restrict__ scal, const double x_cost,
                                                              const double min_relative_decrease, const int enabled, const int armijo,
                                                              const Publish pb)
{
    if (threadIdx.x == 0) {
        const double mcc = scal[SC_MCC_PTS] + scal[SC_MCC_CAM], cand = scal[SC_CAND_COST], gdot = scal[SC_GDOT_PTS] + scal[SC_GDOT_CAM];
        const bool ok = scal[SC_LIN_STATUS] == 0.0 && scal[SC_SCHUR_STATUS] == 0.0 && isfinite(mcc) && isfinite(cand) && mcc > 0.0 &&
                        (!armijo || cand <= x_cost + 1e-4 * gdot);
        scal[SC_GATE] = (enabled && ok && (x_cost - cand) / mcc > min_relative_decrease) ? 1.0 : 0.0;
    }
    publish_tail(pb);
}

// Camera block and gradient of one camera from the tile partials, by one 64-lane workgroup: lanes 0..26 each own one value
// and add the camera's tiles in order (bitwise reproducible).  The sums arrive without the camera's constant M (see
// ba_linearize): B = T^T B~ T, g = T^T g~ with T = blockdiag(M, I) is applied here, once per camera instead of once per
// observation.  Out (LDS): Bc[36] (both triangles from the same expression), gc6[6].  bt / gt / Tm: LDS scratch.
__device__ __forceinline__ void cam_block_from_tiles(const uint32_t cam, const uint32_t* __restrict__ cam_tile_start,
                                                     const double* __restrict__ tile_part, const double* __restrict__ campre,
                                                     double* __restrict__ bt, double* __restrict__ gt, double* __restrict__ Tm,
                                                     double* __restrict__ Bc, double* __restrict__ gc6)
{
    const int v = threadIdx.x;
    if (v < 27) {
        double s = 0.0;
        for (uint32_t t = cam_tile_start[cam]; t < cam_tile_start[cam + 1]; t++) s += tile_part[(size_t)t * kTileVals + v];
        if (v < 21) {
            int a = 0, rem = v;
            while (rem >= 6 - a) { rem -= 6 - a; a++; }
            const int b = a + rem;
            bt[a * 6 + b] = s;
            bt[b * 6 + a] = s;
        } else {
            gt[v - 21] = s;
        }
    }
    if (v < 36) {
        const int i = v / 6, j = v % 6;
        Tm[v] = (i < 3 && j < 3) ? campre[kPoseStride * (size_t)cam + 9 + i * 3 + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    if (v < 36) {
        // B[a][b] = sum_ij T[i][a] B~[i][j] T[j][b], fixed order; the lower triangle mirrors the upper one
        const int a0 = v / 6, b0 = v % 6, a = a0 < b0 ? a0 : b0, b = a0 < b0 ? b0 : a0;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double u = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) u += bt[i * 6 + j] * Tm[j * 6 + b];
            s += Tm[i * 6 + a] * u;
        }
        Bc[v] = s;
    } else if (v < 42) {
        const int a = v - 36;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) s += Tm[i * 6 + a] * gt[i];
        gc6[a] = s;
    }
    __syncthreads();
}

// K4  per-point J_p^T J_p and J_p^T r through the point-major index; one lane per point, each
// observation a 128-B row.
__global__ __launch_bounds__(kPointBlock) void ba_point_reduce_kernel(uint32_t n_pt, const uint32_t* __restrict__ pt_start,
                                                                      const uint32_t* __restrict__ pt_obs,
                                                                      const uint32_t* __restrict__ q_cam,
                                                                      const CompactRows ar,
                                                                      const double* __restrict__ campre,
                                                                      double* __restrict__ C, double* __restrict__ gp,
    const double* __restrict__ gate)
{
    if (gate && *gate == 0.0) return;   // speculative launch (see ba_step_sums_kernel): the step was not accepted
    // kBacksubLanes lanes per point share its observations (a gather: more rows in flight), fixed butterfly at the end.
    // J_p = A R is rebuilt from the compact row and the camera's rotation (cached: 72 B per camera).
    const uint32_t p = (blockIdx.x * kPointBlock + threadIdx.x) / kBacksubLanes;
    const uint32_t sub = threadIdx.x % kBacksubLanes;
    const bool live = p < n_pt;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, g0 = 0, g1 = 0, g2 = 0;
    if (live) {
        for (uint32_t q = pt_start[p] + sub; q < pt_start[p + 1]; q += kBacksubLanes) {
            const size_t k = pt_obs[q];
            const double2* row = reinterpret_cast<const double2*>(ar.g + kArG * k);
            const double2* hrow = reinterpret_cast<const double2*>(ar.h + kArH * k);
            const double* Rc = campre + kPoseStride * (size_t)q_cam[q];
            double w[10], R[10], z[9];
#pragma unroll
            for (int i = 0; i < 3; i++) { const double2 d = row[i]; w[2 * i] = d.x; w[2 * i + 1] = d.y; }
#pragma unroll
            for (int i = 0; i < 2; i++) { const double2 d = hrow[i]; w[6 + 2 * i] = d.x; w[7 + 2 * i] = d.y; }
#pragma unroll
            for (int i = 0; i < 5; i++) { const double2 d = reinterpret_cast<const double2*>(Rc)[i]; R[2 * i] = d.x; R[2 * i + 1] = d.y; }
            compact_gr(w, R, z);                       // Z = G R;  J_p^T J_p = R^T Z,  J_p^T r = R^T h
            c0 += R[0] * z[0] + R[3] * z[3] + R[6] * z[6];
            c1 += R[0] * z[1] + R[3] * z[4] + R[6] * z[7];
            c2 += R[0] * z[2] + R[3] * z[5] + R[6] * z[8];
            c3 += R[1] * z[1] + R[4] * z[4] + R[7] * z[7];
            c4 += R[1] * z[2] + R[4] * z[5] + R[7] * z[8];
            c5 += R[2] * z[2] + R[5] * z[5] + R[8] * z[8];
            g0 += R[0] * w[6] + R[3] * w[7] + R[6] * w[8];
            g1 += R[1] * w[6] + R[4] * w[7] + R[7] * w[8];
            g2 += R[2] * w[6] + R[5] * w[7] + R[8] * w[8];
        }
    }
    c0 = group_sum<kBacksubLanes>(c0); c1 = group_sum<kBacksubLanes>(c1); c2 = group_sum<kBacksubLanes>(c2);
    c3 = group_sum<kBacksubLanes>(c3); c4 = group_sum<kBacksubLanes>(c4); c5 = group_sum<kBacksubLanes>(c5);
    g0 = group_sum<kBacksubLanes>(g0); g1 = group_sum<kBacksubLanes>(g1); g2 = group_sum<kBacksubLanes>(g2);
    if (!live || sub != 0) return;
    double* Cp = C + 6 * (size_t)p;
    Cp[0] = c0; Cp[1] = c1; Cp[2] = c2; Cp[3] = c3; Cp[4] = c4; Cp[5] = c5;
    double* g = gp + 3 * (size_t)p;
    g[0] = g0; g[1] = g1; g[2] = g2;
}

__global__ __launch_bounds__(kPointBlock) void ba_point_scale_kernel(uint32_t n_pt, const double* __restrict__ C, int jacobi,
                                                                     double* __restrict__ sp)
{
    const uint32_t p = blockIdx.x * kPointBlock + threadIdx.x;
    if (p >= n_pt) return;
    const double* Cp = C + 6 * (size_t)p;
    sp[3 * (size_t)p] = jacobi ? 1.0 / (1.0 + sqrt(Cp[0])) : 1.0;
    sp[3 * (size_t)p + 1] = jacobi ? 1.0 / (1.0 + sqrt(Cp[3])) : 1.0;
    sp[3 * (size_t)p + 2] = jacobi ? 1.0 / (1.0 + sqrt(Cp[5])) : 1.0;
}

// ---------------------------------------------------------------------------------------------------
// K5  ba_schur: one workgroup per chunk of consecutive points whose free cameras fit a window of KMAX local
// slots.  With the damped point block C_p = L L^T (3x3 Cholesky) and U~_p = W~_p L^-T, the window's 6 KMAX x 6 KMAX matrix
//     sum_p W~_p C_p^-1 W~_p^T = sum_p U~_p U~_p^T
// (U~_p the point's 6 KMAX x 3 column block, zero rows for cameras that do not see it) is a SYMMETRIC rank-k update over
// the concatenated point columns, and it runs on the f64 matrix cores out of ONE LDS image (both MFMA operands are tile
// rows of the same image; half the LDS of the two-image form W~ C^-1 . W~^T, so three workgroups share a CU).  W~ is
// J_c^T J_p without the per-camera factor T = blockdiag(M^T, I) (ba_device.h): ba_schur_reduce applies T once per block.
// Per batch (<= 128 observations, <= PB points):
//   * all lanes fetch the G part (48 B) of the batch's compact rows from HBM as 16-byte pieces - requested a whole
//     batch AHEAD, into registers - and park them in LDS;
//   * one lane per point factors the damped 3x3 block of the NEXT batch's points in registers (L^-T, t = L^-1 g and
//     C^-1 = L^-T L^-1 for the back-substitution); two lanes per observation form W~ from G, the camera's rotation (an
//     LDS table of the chunk's window cameras) and the point, and write U~ = W~ L^-T into the zero-filled LDS image laid
//     out [point column][window row];
//   * every wave owns a fixed set of 16x16 tiles of the window's upper triangle and accumulates
//     v_mfma_f64_16x16x4_f64 products over the batch's columns in registers (no atomics, fixed order); lane groups
//     add the rhs part  U~ t = W~ C^-1 g  with plain FMAs.
// The chunk's window goes to its own slab ([pair a <= b][6x6] then [camera][6]); ba_schur_reduce sums the slabs
// per block in a fixed order, so the reduced camera system is bitwise reproducible.
// ---------------------------------------------------------------------------------------------------
constexpr int kRawRow = 6;                        // G: the first 48 B of the compact row staged per observation
constexpr int kRawPieces = kRawRow / 2;           // 16-byte pieces per row
// Workgroup size by window width.  Up to 16 cameras: 256 lanes and batches of 128 observations, so that several
// workgroups share a CU (40 KB of LDS each) and one runs while another waits at its barriers; wider windows keep 512
// lanes - their 78 accumulator tiles need the registers of eight waves.
constexpr int schur_threads(int kmax) { return kmax <= 16 ? 256 : 512; }
constexpr int kCamTab = 10;                       // per window camera: R (9), first-order-branch flag
constexpr int kPointVals = 12;                    // per point and batch: M = L^-T (6), t = L^-1 J_p^T r (3), position (3)
constexpr int kPtFac = 12;                        // per point in global memory: M (6), t (3), position (3) - 96 B, six 16-byte
                                                  // pieces; written by the chunk's prologue

typedef double schur_double4 __attribute__((ext_vector_type(4)));

template <int KMAX>
struct SchurShape {
    static constexpr int PB = schur_batch_points(KMAX);   // points per batch
    static constexpr int ROWS = 6 * KMAX;                  // window rows
    // leading dimension of the image: the matrix-core operand reads take 16 consecutive doubles of 4 consecutive point
    // columns per instruction (lanes 0-15 column k, 16-31 column k + 1 in one bank group): with LD = 16 (mod 32) doubles
    // two consecutive columns lie exactly half the banks apart - conflict-free
    static constexpr int LD = ((ROWS + 15) / 32) * 32 + 16;
    static constexpr int KB = 3 * PB;                      // point columns per batch (multiple of 4)
    static constexpr int NT1 = (ROWS + 15) / 16;           // 16-wide tiles per dimension
    static constexpr int NUP = NT1 * (NT1 + 1) / 2;        // tiles of the upper triangle
    static constexpr int NT = schur_threads(KMAX);         // lanes per workgroup
    static constexpr int TPW = (NUP + NT / 64 - 1) / (NT / 64);   // tiles per wave
    static constexpr size_t lds_bytes =
        sizeof(double) * ((size_t)kBatchObs * kRawRow + (size_t)KB * LD + 2 * (PB * 6 + KB + PB * 3) + KMAX * kCamTab) + 2 * kBatchObs;
    static_assert(LD >= ROWS && LD % 32 == 16, "image leading dimension");
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load
// (s_waitcnt vmcnt(0)), which would drain the next batch's prefetch at each barrier; the prefetched registers are
// guarded by the compiler's own vmcnt bookkeeping at their first use.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The window product of one batch for a wave that owns the first NA of its tile slots: KB / 4 matrix instructions per
// tile, k outermost (the tiles are independent accumulators, their products issue back to back).
template <int NA, int TPW, int KB, int LD>
__device__ __forceinline__ void schur_mfma_body(const double* const (&pa)[TPW], const double* const (&pb)[TPW], schur_double4 (&acc)[TPW])
{
    if constexpr (TPW <= 6) {
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 4) {
            double av[NA], bv[NA];
#pragma unroll
            for (int j = 0; j < NA; j++) { av[j] = pa[j][k0 * LD]; bv[j] = pb[j][k0 * LD]; }
#pragma unroll
            for (int j = 0; j < NA; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc[j], 0, 0, 0);
        }
    } else {   // wide windows: ten tiles per wave leave no registers for an unrolled operand stream
#pragma unroll 1
        for (int k0 = 0; k0 < KB; k0 += 4) {
            double av[NA], bv[NA];
#pragma unroll
            for (int j = 0; j < NA; j++) { av[j] = pa[j][k0 * LD]; bv[j] = pb[j][k0 * LD]; }
#pragma unroll
            for (int j = 0; j < NA; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc[j], 0, 0, 0);
        }
    }
}

template <int TPW, int KB, int LD, int NA = TPW>
__device__ __forceinline__ void schur_mfma_dispatch(int n_act, const double* const (&pa)[TPW], const double* const (&pb)[TPW],
                                                    schur_double4 (&acc)[TPW])
{
    if constexpr (TPW > 6) {
        // wide windows (ten tile slots per wave): one rolled loop with a wave-uniform test per tile - a body per tile
        // count would cost more registers than the kernel has
#pragma unroll 1
        for (int k0 = 0; k0 < KB; k0 += 4) {
            double av[TPW], bv[TPW];
#pragma unroll
            for (int j = 0; j < TPW; j++) { av[j] = pa[j][k0 * LD]; bv[j] = pb[j][k0 * LD]; }
#pragma unroll
            for (int j = 0; j < TPW; j++)
                if (j < n_act) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc[j], 0, 0, 0);
        }
    } else if constexpr (NA >= 1) {
        if (n_act == NA) schur_mfma_body<NA, TPW, KB, LD>(pa, pb, acc);
        else schur_mfma_dispatch<TPW, KB, LD, NA - 1>(n_act, pa, pb, acc);
    }
}

// Four tile rows on four waves (windows of up to ten cameras - the reference's usual track length; ba_schur10_kernel): the ten tiles are dealt
// so that a wave's tiles share operand blocks, and each 16-column block of the image is read ONCE per k step and wave and
// used in every product it appears in (an operand block is the same data whether it feeds A or B):
//   wave 0: (0,0) (0,1) (0,2)   blocks 0 1 2        wave 2: (2,2) (2,3)   blocks 2 3
//   wave 1: (1,1) (1,2) (1,3)   blocks 1 2 3        wave 3: (0,3) (3,3)   blocks 0 3
// ten block reads per k step instead of twenty - the kernel is bound by LDS traffic (182 KB per batch and workgroup, half
// of it operand reads), not by the matrix pipe.
template <int KB, int LD>
__device__ __forceinline__ void schur_mfma_shared4(const int wave, const double* __restrict__ x0p, const double* __restrict__ x1p,
                                                   const double* __restrict__ x2p, schur_double4 (&acc)[3])
{
    if (wave < 2) {
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 4) {
            const double x0 = x0p[k0 * LD], x1 = x1p[k0 * LD], x2 = x2p[k0 * LD];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x2, acc[2], 0, 0, 0);
        }
    } else if (wave == 2) {
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 4) {
            const double x0 = x0p[k0 * LD], x1 = x1p[k0 * LD];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, acc[1], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int k0 = 0; k0 < KB; k0 += 4) {
            const double x0 = x0p[k0 * LD], x1 = x1p[k0 * LD];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc[1], 0, 0, 0);
        }
    }
}

template <int KMAX>
__global__ __launch_bounds__(schur_threads(KMAX), KMAX <= 10 ? 3 : 2) void ba_schur_kernel(
    const SchurChunk* __restrict__ chunks, const SchurBatch* __restrict__ batches, const uint32_t* __restrict__ chunk_slab,
    const uint32_t* __restrict__ chunk_cam, const uint32_t* __restrict__ pt_obs, const uint32_t* __restrict__ q_pt,
    const uint8_t* __restrict__ q_slot, const double* __restrict__ ar, const double* __restrict__ campre,
    const double* __restrict__ pts, const double* __restrict__ C, const double* __restrict__ gp,
    const double* __restrict__ sp, const LmDiag lm, double* __restrict__ Cinv, double* __restrict__ ptfac,
    double* __restrict__ slab, double* __restrict__ scal)
{
    using Sh = SchurShape<KMAX>;
    constexpr int NT = Sh::NT, PB = Sh::PB, LD = Sh::LD, KB = Sh::KB, NT1 = Sh::NT1, NUP = Sh::NUP, TPW = Sh::TPW;
    constexpr int NPIECE = kBatchObs * kRawPieces;                // 16-byte pieces per batch
    constexpr int PPT = (NPIECE + NT - 1) / NT;                   // per lane
    static_assert(NT >= 2 * kBatchObs, "W/Y staging uses two lanes per observation");
    static_assert(KB % 4 == 0 && (KB * LD) % 2 == 0, "image shape");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* const raw = lds;                              // [kBatchObs][6]
    double* const Ui = raw + kBatchObs * kRawRow;         // [KB][LD]   Ui[3 pl + c][6 slot + r] = U~[r][c],  U~ = W~ L^-T
    double* const cil = Ui + KB * LD;                     // [2][PB][6] L^-T of the damped point blocks (m00 m01 m02 m11 m12 m22), by batch parity
    double* const gl = cil + 2 * PB * 6;                  // [2][KB]    t = L^-1 J_p^T r of the batch's points, index 3 pl + c
    double* const xl = gl + 2 * KB;                       // [2][PB][3] the batch's points (linearisation point)
    double* const camtab = xl + 2 * PB * 3;               // [KMAX][10] R and branch flag of the window's cameras
    uint8_t* const pt_l = reinterpret_cast<uint8_t*>(camtab + KMAX * kCamTab);   // batch-local point of each staged observation
    uint8_t* const slot_l = pt_l + kBatchObs;                                    // its window slot (255: fixed camera)

    const SchurChunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid % 64;
    const int wave = __builtin_amdgcn_readfirstlane(tid / 64);
    const int K = (int)ch.n_local, rows_used = 6 * K;
    const int ptl = tid - (NT - 64);   // point lane of the last wave (negative elsewhere)

    // The chunk's batch descriptors live in registers, lane l of every wave holding batch l: the three descriptors a loop
    // iteration needs (this batch, the next one's rows, the one after's indices) are lane reads, not scalar loads with a
    // wait for the scalar cache in front of each (about 300 cycles apiece, three times per batch).
    const uint32_t n_batch = ch.batch_end - ch.batch_begin;
    SchurBatch bd_mine = n_batch ? batches[ch.batch_begin + ((uint32_t)lane < n_batch ? (uint32_t)lane : n_batch - 1)] : SchurBatch{0, 0, 0, 0, 0};
    auto batch_at = [&](uint32_t bi) __attribute__((always_inline)) -> SchurBatch {
        const uint32_t r = bi - ch.batch_begin;   // wave-uniform
        if (r >= 64) return batches[bi];
        SchurBatch b;
        b.q_begin = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.q_begin, (int)r);
        b.q_end = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.q_end, (int)r);
        b.p_begin = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.p_begin, (int)r);
        b.p_end = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.p_end, (int)r);
        b.full = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.full, (int)r);
        return b;
    };

    // the window cameras' rotations, once per chunk (read by the W/Y lanes after the first barrier of the loop)
    for (int e = tid; e < K * kCamTab; e += NT) {
        const int slot = e / kCamTab, i = e - slot * kCamTab;
        camtab[e] = campre[kPoseStride * (size_t)chunk_cam[(size_t)blockIdx.x * KMAX + slot] + (i < 9 ? i : 21)];
    }

    // The chunk's points, all at once, one lane each: Cholesky factor of the damped 3x3 block, M = L^-T, t = L^-1 g = M^T g and
    // C^-1 = M M^T.  A long dependent f64 chain (three square roots, the divisions behind them) - taken once per chunk with
    // every lane busy, not once per batch on twelve lanes of one wave with the other waves waiting at the barrier for it
    // (measured: 2 550 of the 9 000 cycles of a batch).  M and t go to global memory (72 B per point, read back a batch
    // ahead by the point lanes), C^-1 is an output anyway (ba_backsub).
    for (uint32_t idx = tid; idx < (ch.batch_end - ch.batch_begin) * (uint32_t)PB; idx += NT) {
        const SchurBatch bt = batches[ch.batch_begin + idx / PB];
        const uint32_t pl = idx % PB;
        if (pl >= bt.p_end - bt.p_begin) continue;
        const size_t p = (size_t)bt.p_begin + pl;
        double c[6], mt[6], ci[6];
#pragma unroll
        for (int i = 0; i < 6; i++) c[i] = C[6 * p + i];
        const double g0 = gp[3 * p], g1 = gp[3 * p + 1], g2 = gp[3 * p + 2];
        const double m[6] = {c[0] + point_lambda(c[0], sp[3 * p], lm), c[1], c[2], c[3] + point_lambda(c[3], sp[3 * p + 1], lm), c[4],
                             c[5] + point_lambda(c[5], sp[3 * p + 2], lm)};
        if (!sym3_chol_inverse(m, mt, ci)) scal[SC_SCHUR_STATUS] = 1.0;
        double* f = ptfac + kPtFac * p;
        double* o = Cinv + 6 * p;
#pragma unroll
        for (int i = 0; i < 6; i++) { f[i] = mt[i]; o[i] = ci[i]; }
        f[6] = mt[0] * g0;
        f[7] = mt[1] * g0 + mt[3] * g1;
        f[8] = mt[2] * g0 + mt[4] * g1 + mt[5] * g2;
    }
    // the point lanes of this workgroup read M and t back: same CU, same L1 - workgroup scope is all the ordering needed (an
    // agent-scope fence here writes the whole L2 back, once per workgroup: measured 75 -> 150 us)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // this wave's tiles of the upper triangle of the rows the chunk USES (a window of 11 cameras covers 5 of the 6 tile
    // rows a 16-camera window has): the used tiles u = wave, wave + NT/64, .. are dealt round the waves, so the matrix
    // work stays balanced whatever the window's width
    int t_i0[TPW], t_j0[TPW];
    schur_double4 acc[TPW];
    const int nt1_used = (rows_used + 15) / 16, nup_used = nt1_used * (nt1_used + 1) / 2;
    static_assert(TPW * (NT / 64) >= NUP && NT1 * 16 >= 6 * KMAX, "every tile of the widest window has an owner");
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        int u = wave + j * (NT / 64), ti = 0;
        t_i0[j] = -1; t_j0[j] = 0;
        if (u < nup_used) {
            while (u >= nt1_used - ti) { u -= nt1_used - ti; ti++; }
            t_i0[j] = ti * 16; t_j0[j] = (ti + u) * 16;
        }
        acc[j] = schur_double4{0.0, 0.0, 0.0, 0.0};
    }
    // operand bases of the wave's tiles in the image (fixed for the chunk) and the number of tiles it owns
    const double* pa[TPW];
    const double* pb[TPW];
    int n_act = 0;
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        const int i0 = t_i0[j] < 0 ? 0 : t_i0[j];
        pa[j] = Ui + (lane / 16) * LD + i0 + lane % 16;
        pb[j] = Ui + (lane / 16) * LD + t_j0[j] + lane % 16;
        n_act += t_i0[j] >= 0 ? 1 : 0;
    }
    // The rhs product U~ t belongs to one lane per window row, in the waves just below the point wave: those own the fewest
    // tiles (the tiles are dealt from wave 0 up), so the product fills their idle time instead of lengthening the waves
    // every barrier waits for.
    constexpr int kRhsFirst = NT - 64 - ((Sh::ROWS + 63) / 64) * 64;
    static_assert(kRhsFirst >= 0, "room for one rhs lane per window row below the point wave");
    const int rrow = tid - kRhsFirst;   // this lane's row of the rhs (outside [0, rows_used): none)
    double racc = 0.0;
    // ... unless the window leaves a spare column in its last tile column (6 K is not a multiple of 16: every K but 8, 16,
    // 24, 32): then t sits in image column 6 K, and the tiles of that column deliver U~ t in their column 6 K for nothing
    const bool rhs_in_tiles = rows_used % 16 != 0;

    // Prefetch registers, one batch ahead: this lane's 16-byte pieces of the G part of the batch's compact rows; its
    // observation's point id and window slot; and - one lane per point - the point's J_p^T J_p, Jacobi scale, J_p^T r
    // and position.  Everything that does not depend on the batch is worked out once, here; row offsets are 32-bit
    // (the host refuses problems whose row array exceeds 4 GiB).
    double pre_x[PPT], pre_y[PPT];
    uint32_t pre_row[PPT];     // camera-major row of each piece's observation, fetched TWO batches ahead
    uint32_t pre_pt = 0;
    uint8_t pre_slot = 255;
    // the last wave carries the next batch's point data, three values per lane (15 per point: see kPointVals); the point
    // lanes collect theirs through LDS inside that wave - 6 registers instead of 30 in every lane of the workgroup
    constexpr int kPtPerLane = (PB * kPointVals + 63) / 64;
    double pre_pv[kPtPerLane];
#pragma unroll
    for (int u = 0; u < kPtPerLane; u++) pre_pv[u] = 0.0;
    uint32_t p_pack[PPT];      // pass s: batch-local observation of this lane's piece | piece inside the row << 16
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        const int piece = tid + s * NT;
        const int o = piece / kRawPieces, part = piece - o * kRawPieces;   // o >= kBatchObs in the idle tail of the last pass
        p_pack[s] = (uint32_t)o | ((uint32_t)part << 16);
        pre_x[s] = 0.0; pre_y[s] = 0.0; pre_row[s] = 0;
    }
    auto p_obs = [&](int s) __attribute__((always_inline)) { return (int)(p_pack[s] & 0xFFFFu); };
    auto p_part = [&](int s) __attribute__((always_inline)) { return p_pack[s] >> 16; };
    const char* const ar_b = reinterpret_cast<const char*>(ar);
    auto fetch_index = [&](uint32_t bi) __attribute__((always_inline)) {
        const SchurBatch bt = batch_at(bi);
        const int nq = (int)(bt.q_end - bt.q_begin);
#pragma unroll
        for (int s = 0; s < PPT; s++) pre_row[s] = p_obs(s) < nq ? pt_obs[bt.q_begin + p_obs(s)] : 0u;
    };
    auto fetch = [&](uint32_t bi) __attribute__((always_inline)) {   // rows of batch bi through the indices fetch_index(bi) loaded earlier
        const SchurBatch bt = batch_at(bi);
        const int nq = (int)(bt.q_end - bt.q_begin), np = (int)(bt.p_end - bt.p_begin);
#pragma unroll
        for (int s = 0; s < PPT; s++) {
            // unconditional: idle lanes carry row 0 from fetch_index and read a valid (cached) piece they never store
            const uint32_t off = pre_row[s] * (uint32_t)(kArG * 8) + p_part(s) * 16u;
            const double2 v = *reinterpret_cast<const double2*>(ar_b + off);
            pre_x[s] = v.x; pre_y[s] = v.y;
        }
        if (tid < nq) { pre_pt = q_pt[bt.q_begin + tid]; pre_slot = q_slot[bt.q_begin + tid]; }
        if (ptl >= 0) {
#pragma unroll
            for (int u = 0; u < kPtPerLane; u++) {
                const int idx = ptl * kPtPerLane + u, pl = idx / kPointVals, v = idx - pl * kPointVals;
                const size_t p = (size_t)bt.p_begin + pl;
                double val = 0.0;
                if (pl < np) val = v < 9 ? ptfac[kPtFac * p + v] : pts[3 * p + (v - 9)];
                pre_pv[u] = val;
            }
        }
    };
    // The point lanes (the last wave) hand the batch AFTER the one being multiplied its M, t and positions: plain copies into
    // the parity buffers of that batch; lanes past the batch's points carry zeros (the k padding of the products).
    auto point_phase = [&](uint32_t bi) __attribute__((always_inline)) {
        if (ptl < 0) return;
        double* cl = cil + (bi & 1) * (PB * 6);
        double* g = gl + (bi & 1) * KB;
        double* xp = xl + (bi & 1) * (PB * 3);
#pragma unroll
        for (int u = 0; u < kPtPerLane; u++) {
            const int idx = ptl * kPtPerLane + u, pl = idx / kPointVals, v = idx - pl * kPointVals;
            if (idx >= PB * kPointVals) continue;
            double* dst = v < 6 ? cl + pl * 6 + v : v < 9 ? g + pl * 3 + (v - 6) : xp + pl * 3 + (v - 9);
            *dst = pre_pv[u];
        }
    };
    if (ch.batch_begin < ch.batch_end) {
        fetch_index(ch.batch_begin);
        fetch(ch.batch_begin);
        if (ch.batch_begin + 1 < ch.batch_end) fetch_index(ch.batch_begin + 1);
        point_phase(ch.batch_begin);
    }

    for (uint32_t bi = ch.batch_begin; bi < ch.batch_end; bi++) {
        const SchurBatch bt = batch_at(bi);
        const int nq = (int)(bt.q_end - bt.q_begin);
        lds_barrier();   // previous batch's products are done with the LDS images
#pragma unroll
        for (int s = 0; s < PPT; s++)
            if (p_obs(s) < nq) reinterpret_cast<double2*>(raw)[p_obs(s) * kRawPieces + (int)p_part(s)] = make_double2(pre_x[s], pre_y[s]);
        const uint32_t my_pt = pre_pt;
        const uint8_t my_slot = pre_slot;
        // the prefetch registers are free again: request the next batch now, a whole batch time ahead of its use
        if (bi + 1 < ch.batch_end) {
            fetch(bi + 1);
            if (bi + 2 < ch.batch_end) fetch_index(bi + 2);
        }
        // zero the image (contiguous, 16-byte stores); absent cameras and the k padding stay zero.  Not needed when the batch
        // overwrites every element the products read (bt.full, set by the host): what is left over in the padding columns
        // then only reaches accumulator columns that are never stored.
        if (!bt.full)
            for (int e = tid; e < KB * LD / 2; e += NT) reinterpret_cast<double2*>(Ui)[e] = make_double2(0.0, 0.0);
        if (tid < nq) { pt_l[tid] = (uint8_t)(my_pt - bt.p_begin); slot_l[tid] = my_slot; }
        const double* const cil_b = cil + (bi & 1) * (PB * 6);
        const double* const gl_b = gl + (bi & 1) * KB;
        const double* const xl_b = xl + (bi & 1) * (PB * 3);
        lds_barrier();
        // U~ = W~ L^-T: two lanes per observation, the rotation rows (0..2) and the translation rows (3..5)
        if (tid < 2 * nq) {
            const int o = tid >> 1, half = tid & 1;
            const int slot = slot_l[o];
            if (slot != 255) {
                const double* Gr = raw + o * kRawRow;
                const int pl = pt_l[o];
                const double* mi = cil_b + pl * 6;
                const double* ct = camtab + slot * kCamTab;
                double G[6], R[9], v[3], w[18];
#pragma unroll
                for (int i = 0; i < 6; i++) G[i] = Gr[i];
#pragma unroll
                for (int i = 0; i < 9; i++) R[i] = ct[i];
                compact_v(R, ct[9] != 0.0, xl_b + pl * 3, v);
                compact_wt(G, R, v, w);
                const double m00 = mi[0], m01 = mi[1], m02 = mi[2], m11 = mi[3], m12 = mi[4], m22 = mi[5];
                double* Uc = Ui + (pl * 3) * LD + slot * 6;
#pragma unroll
                for (int rr = 0; rr < 3; rr++) {
                    const int r = half * 3 + rr;
                    const double w0 = half ? w[(3 + rr) * 3] : w[rr * 3], w1 = half ? w[(3 + rr) * 3 + 1] : w[rr * 3 + 1],
                                 w2 = half ? w[(3 + rr) * 3 + 2] : w[rr * 3 + 2];
                    Uc[r] = w0 * m00;
                    Uc[LD + r] = w0 * m01 + w1 * m11;
                    Uc[2 * LD + r] = w0 * m02 + w1 * m12 + w2 * m22;
                }
            }
        }
        if (rhs_in_tiles && tid < KB) Ui[tid * LD + rows_used] = gl_b[tid];
        lds_barrier();
        // window += Ui^T Ui over the batch's columns: lane l feeds A[i = l%16][k = l/16] and B[k = l/16][j = l%16], both
        // from the one image.  Straight-line code: the wave's n_act tiles (wave-uniform) pick one fully unrolled body, all
        // KB / 4 steps are taken (the k padding of a short batch is zero), every operand read is base + immediate offset
        schur_mfma_dispatch<TPW, KB, LD>(n_act, pa, pb, acc);
        // rhs part U~ t: all KB columns, straight-line (the k padding of a short batch is zero in both factors), every operand
        // requested before the first product
        if (!rhs_in_tiles && rrow >= 0 && rrow < rows_used) {
            const double* y = Ui + rrow;
            constexpr int kStep = 12;   // operands in flight: 24 doubles (all KB at once would spill)
            static_assert(KB % kStep == 0, "whole steps");
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k0 = 0; k0 < KB; k0 += kStep) {
                double yv[kStep], tv[kStep];
#pragma unroll
                for (int k = 0; k < kStep; k++) { yv[k] = y[(k0 + k) * LD]; tv[k] = gl_b[k0 + k]; }
#pragma unroll
                for (int k = 0; k < kStep; k += 2) { s0 += yv[k] * tv[k]; s1 += yv[k + 1] * tv[k + 1]; }
            }
            racc += s0 + s1;
        }
        if (bi + 1 < ch.batch_end) point_phase(bi + 1);   // its inputs were requested at the top of this iteration
    }

    // the chunk's window: [pair a <= b][6x6] then [camera][6].  Accumulator register r of lane l holds element
    // (i0 + 4 r + l/16, j0 + l%16); the same-camera blocks are symmetric (U~ U~^T), so their lower entries are the
    // mirrored upper ones, wherever the 16x16 tiling cuts them.
    double* out = slab + chunk_slab[blockIdx.x];
    const int n_pair = K * (K + 1) / 2;
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        if (t_i0[j] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = t_i0[j] + 4 * r + lane / 16, col = t_j0[j] + lane % 16;
            if (rhs_in_tiles && col == rows_used && row < rows_used) out[n_pair * 36 + row] = acc[j][r];
            if (row >= rows_used || col >= rows_used) continue;
            const int a = row / 6, rr = row - 6 * a, b = col / 6, cc = col - 6 * b;
            if (a > b) continue;
            const int pidx = a * K - a * (a - 1) / 2 + (b - a);
            double* blk = out + pidx * 36;
            const double v = acc[j][r];
            if (a < b) {
                blk[rr * 6 + cc] = v;
            } else if (row <= col) {
                blk[rr * 6 + cc] = v;
                if (row < col) blk[cc * 6 + rr] = v;
            }
        }
    }
    if (!rhs_in_tiles && rrow >= 0 && rrow < rows_used) out[n_pair * 36 + rrow] = racc;
}

// ---------------------------------------------------------------------------------------------------
// ba_schur10: the windowed elimination for windows of at most ten cameras (the reference's usual track length), software-
// pipelined.  Measured on the kernel above (cycles of one batch, one workgroup, config 3): staging 400, requests 830, zeroing
// 750, W phase 1 600, products 2 200, four barriers and their skew 1 500 - phases in a row, each bound by latency or by LDS
// traffic (182 KB per batch: image zeroing, staged rows, the W lanes' table reads, twenty operand-block reads per k step).
// Here the phases overlap and most of that traffic is gone:
//   * eight waves: four PRODUCERS form the image of batch b + 1 while four CONSUMERS multiply the image of batch b - two
//     images in LDS, ONE barrier per batch;
//   * a producer lane pair owns one (point, window slot) pair of the batch for the whole chunk: its camera's rotation lives
//     in registers (no camera table), its row index comes from a host-built table (pair_row; 0xFFFFFFFF: the point is not
//     seen by that camera - the pair writes zeros, so the image is never zeroed separately), its point's factor M, t and
//     position arrive in registers a batch ahead (no staging buffers, no intra-producer barrier);
//   * consumers read every operand block once per k step (schur_mfma_shared4), all ten tiles whatever the window's width.
// LDS traffic per batch: 17 KB of image writes + 45 KB of operand reads.
// ---------------------------------------------------------------------------------------------------
constexpr int kS10Kmax = 10;
constexpr int kS10PB = 12;                      // points per batch (= schur_batch_points(10))
constexpr int kS10KB = 3 * kS10PB;              // image rows (k)
constexpr int kS10LD = SchurShape<10>::LD;      // 80
constexpr int kS10Threads = 512;
constexpr size_t kS10Lds = sizeof(double) * (2 * (kS10KB * kS10LD + kS10PairsPerBatch * 6 + kS10PB * kPtFac) + kS10Kmax * 9);   // images, G, point blocks, rotations
static_assert(kS10PairsPerBatch == kS10PB * kS10Kmax && 2 * kS10PairsPerBatch <= kS10Threads / 2, "two producer lanes per (point, slot) pair");

__global__ __launch_bounds__(kS10Threads, 4) void ba_schur10_kernel(
    const SchurChunk* __restrict__ chunks, const SchurBatch* __restrict__ batches, const uint32_t* __restrict__ chunk_slab,
    const uint32_t* __restrict__ chunk_cam, const uint32_t* __restrict__ pair_row, const CompactRows ar,
    const double* __restrict__ campre, const double* __restrict__ pts, double* __restrict__ C,
    double* __restrict__ gp, const double* __restrict__ sp, const LmDiag lm, double* __restrict__ Cinv,
    double* __restrict__ ptfac, double* __restrict__ slab, double* __restrict__ scal, const uint32_t* __restrict__ pt_start,
    const uint32_t* __restrict__ pt_obs, const uint32_t* __restrict__ q_cam, const uint8_t* __restrict__ q_slot, const int from_rows)
{
    constexpr int PB = kS10PB, KB = kS10KB, LD = kS10LD, NT = kS10Threads;
    constexpr int kPieces = 3 * kS10PairsPerBatch;         // 16-byte pieces of the G parts of a batch's rows
    constexpr int kPiecePasses = (kPieces + 255) / 256;    // per consumer lane
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* const img = lds;                                       // [2][KB][LD]    images, by batch parity
    double* const gbuf = img + 2 * KB * LD;                        // [2][120][6]    G of every (point, slot) pair (zeros: not observed)
    double* const ptbuf = gbuf + 2 * kS10PairsPerBatch * 6;        // [2][PB][12]    M, t, position of the batch's points
    double* const camtab = ptbuf + 2 * PB * kPtFac;                // [K][9]         rotations of the window's cameras (point pass)
    const SchurChunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid % 64;
    const int wave = __builtin_amdgcn_readfirstlane(tid / 64);
    const int K = (int)ch.n_local, rows_used = 6 * K;
    const int nb = (int)(ch.batch_end - ch.batch_begin);
    if (nb <= 0) return;

    // batch descriptors in registers, lane l of every wave holding batch l
    const SchurBatch bd_mine = batches[ch.batch_begin + (uint32_t)(lane < nb ? lane : nb - 1)];
    auto batch_at = [&](int rel) __attribute__((always_inline)) -> SchurBatch {   // rel: wave-uniform
        if (rel >= 64) return batches[ch.batch_begin + (uint32_t)rel];
        SchurBatch b;
        b.q_begin = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.q_begin, rel);
        b.q_end = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.q_end, rel);
        b.p_begin = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.p_begin, rel);
        b.p_end = (uint32_t)__builtin_amdgcn_readlane((int)bd_mine.p_end, rel);
        b.full = 0;
        return b;
    };

    if (from_rows) {
        for (int e = tid; e < K * 9; e += NT) camtab[e] = campre[kPoseStride * (size_t)chunk_cam[(size_t)blockIdx.x * kS10Kmax + e / 9] + e % 9];
        __syncthreads();
    }
    // The chunk's points, all at once, a lane pair each: J_p^T J_p = sum R^T G R and J_p^T r = sum R^T h over the point's compact
    // rows when from_rows is set (what ba_point_reduce does in a launch of its own - its 19 us are saved whenever the
    // linearisation behind an accepted step needs no point pass for anything else; the rows come back out of the L2 for the
    // pipeline below), else read from C / gp; then M = L^-T, t = L^-1 g, C^-1 (see ba_schur_kernel); the position rides along.
    for (int base = 0; base < 2 * nb * PB; base += NT) {   // uniform trip count: every lane takes part in the lane reads
        const int idx = (base + tid) >> 1, sub = tid & 1;
        const int brel = idx / PB < nb ? idx / PB : nb - 1;
        // the batch's point range: out of a neighbour lane's descriptor registers, not out of memory
        uint32_t b_p0 = (uint32_t)__shfl((int)bd_mine.p_begin, brel & 63, 64);
        uint32_t b_p1 = (uint32_t)__shfl((int)bd_mine.p_end, brel & 63, 64);
        if (brel >= 64) {
            const SchurBatch bt = batches[ch.batch_begin + (uint32_t)brel];
            b_p0 = bt.p_begin; b_p1 = bt.p_end;
        }
        const uint32_t pl = (uint32_t)(idx % PB);
        const bool live = idx < nb * PB && pl < b_p1 - b_p0;
        const size_t p = (size_t)b_p0 + (live ? pl : 0u);
        double c[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        if (from_rows) {
            if (live) {
                // the lane's observations (every second one of the point): all their indices in one go, then the rows three at
                // a time, then the arithmetic - three memory round trips for a ten-camera track; the window cameras'
                // rotations come from the table in LDS (slot 255: a fixed camera, not in the window - from memory)
                constexpr int kIdx = 6, kRows = 3;
                const uint32_t q1 = pt_start[p + 1];
                for (uint32_t q0 = pt_start[p] + (uint32_t)sub; q0 < q1; q0 += 2 * kIdx) {
                    uint32_t rk[kIdx], rs[kIdx];
#pragma unroll
                    for (int u = 0; u < kIdx; u++) {
                        const uint32_t q = q0 + 2 * u;
                        rk[u] = q < q1 ? pt_obs[q] : 0xFFFFFFFFu;
                        rs[u] = q < q1 ? (uint32_t)q_slot[q] | (q_cam[q] << 8) : 0u;
                    }
#pragma unroll
                    for (int u0 = 0; u0 < kIdx; u0 += kRows) {
                        double2 rw[kRows][5];
#pragma unroll
                        for (int u = 0; u < kRows; u++) {
                            const size_t kr = rk[u0 + u] != 0xFFFFFFFFu ? rk[u0 + u] : 0u;
                            const double2* row = reinterpret_cast<const double2*>(ar.g + kArG * kr);
                            const double2* hrow = reinterpret_cast<const double2*>(ar.h + kArH * kr);
#pragma unroll
                            for (int i = 0; i < 3; i++) rw[u][i] = row[i];
#pragma unroll
                            for (int i = 0; i < 2; i++) rw[u][3 + i] = hrow[i];
                        }
#pragma unroll
                        for (int u = 0; u < kRows; u++) {
                            if (rk[u0 + u] == 0xFFFFFFFFu) continue;
                            const uint32_t slot = rs[u0 + u] & 255u;
                            const double* Rp = slot != 255u ? camtab + slot * 9 : campre + kPoseStride * (size_t)(rs[u0 + u] >> 8);
                            double w[10], R[9], z[9];
#pragma unroll
                            for (int i = 0; i < 5; i++) { w[2 * i] = rw[u][i].x; w[2 * i + 1] = rw[u][i].y; }
#pragma unroll
                            for (int i = 0; i < 9; i++) R[i] = Rp[i];
                            compact_gr(w, R, z);                       // Z = G R;  J_p^T J_p = R^T Z,  J_p^T r = R^T h
                            c[0] += R[0] * z[0] + R[3] * z[3] + R[6] * z[6];
                            c[1] += R[0] * z[1] + R[3] * z[4] + R[6] * z[7];
                            c[2] += R[0] * z[2] + R[3] * z[5] + R[6] * z[8];
                            c[3] += R[1] * z[1] + R[4] * z[4] + R[7] * z[7];
                            c[4] += R[1] * z[2] + R[4] * z[5] + R[7] * z[8];
                            c[5] += R[2] * z[2] + R[5] * z[5] + R[8] * z[8];
                            g[0] += R[0] * w[6] + R[3] * w[7] + R[6] * w[8];
                            g[1] += R[1] * w[6] + R[4] * w[7] + R[7] * w[8];
                            g[2] += R[2] * w[6] + R[5] * w[7] + R[8] * w[8];
                        }
                    }
                }
            }
            // the pair's halves (every lane takes part; lanes without a point carry zeros)
#pragma unroll
            for (int i = 0; i < 6; i++) c[i] += mov_dpp_f64<kDppXor1>(c[i]);
#pragma unroll
            for (int i = 0; i < 3; i++) g[i] += mov_dpp_f64<kDppXor1>(g[i]);
        } else if (live) {
#pragma unroll
            for (int i = 0; i < 6; i++) c[i] = C[6 * p + i];
#pragma unroll
            for (int i = 0; i < 3; i++) g[i] = gp[3 * p + i];
        }
        if (!live || sub != 0) continue;
        if (from_rows) {
#pragma unroll
            for (int i = 0; i < 6; i++) C[6 * p + i] = c[i];
#pragma unroll
            for (int i = 0; i < 3; i++) gp[3 * p + i] = g[i];
        }
        const double g0 = g[0], g1 = g[1], g2 = g[2];
        const double x0 = pts[3 * p], x1 = pts[3 * p + 1], x2 = pts[3 * p + 2];
        double mt[6], ci[6];
        const double m[6] = {c[0] + point_lambda(c[0], sp[3 * p], lm), c[1], c[2], c[3] + point_lambda(c[3], sp[3 * p + 1], lm), c[4],
                             c[5] + point_lambda(c[5], sp[3 * p + 2], lm)};
        if (!sym3_chol_inverse(m, mt, ci)) scal[SC_SCHUR_STATUS] = 1.0;
        double* f = ptfac + kPtFac * p;
        double* o = Cinv + 6 * p;
#pragma unroll
        for (int i = 0; i < 6; i++) { f[i] = mt[i]; o[i] = ci[i]; }
        f[6] = mt[0] * g0;
        f[7] = mt[1] * g0 + mt[3] * g1;
        f[8] = mt[2] * g0 + mt[4] * g1 + mt[5] * g2;
        f[9] = x0; f[10] = x1; f[11] = x2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // same CU, same L1: workgroup scope orders the read-back
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // Loop iteration b = -2 .. nb - 1, one barrier each:
    //   consumers  request the rows and the point block of batch b + 2 (indices requested an iteration earlier), multiply the
    //              image of batch b, then move what arrived to gbuf / ptbuf[b & 1] - the memory latency hides behind the
    //              products, and the waves that are short of work carry the requests;
    //   producers  form the image of batch b + 1 from gbuf / ptbuf[(b + 1) & 1]: LDS in, LDS out, no memory instruction.
    if (wave < 4) {
        // ---- producers --------------------------------------------------------------------------------------------
        const int e = tid >> 1, half = tid & 1;            // the lane pair's (point, slot) pair; rotation rows / translation rows
        const bool on = e < PB * K;
        const int pl = on ? e / K : 0, slot = on ? e - pl * K : 0;
        double R[9];
        bool small;
        {
            const double* pc = campre + kPoseStride * (size_t)chunk_cam[(size_t)blockIdx.x * kS10Kmax + slot];
#pragma unroll
            for (int i = 0; i < 9; i++) R[i] = pc[i];
            small = pc[21] != 0.0;
        }
        for (int b = -2; b < nb; b++) {
            if (on && b + 1 >= 0 && b + 1 < nb) {
                const int par = (b + 1) & 1;
                const double* pb = ptbuf + par * (PB * kPtFac) + pl * kPtFac;
                const double* gs = gbuf + par * (kS10PairsPerBatch * 6) + e * 6;
                double* im = img + par * (KB * LD);
                double G[6], M[6], x[3], v[3], w[18];
#pragma unroll
                for (int i = 0; i < 6; i++) { G[i] = gs[i]; M[i] = pb[i]; }
#pragma unroll
                for (int i = 0; i < 3; i++) x[i] = pb[9 + i];
                compact_v(R, small, x, v);
                compact_wt(G, R, v, w);
                // U~ = W~ L^-T: rows 3 pl + c (k), columns 6 slot + 3 half + rr; t into column 6 K
                double* Uc = im + (pl * 3) * LD + slot * 6 + half * 3;
#pragma unroll
                for (int rr = 0; rr < 3; rr++) {
                    const double w0 = half ? w[(3 + rr) * 3] : w[rr * 3], w1 = half ? w[(3 + rr) * 3 + 1] : w[rr * 3 + 1],
                                 w2 = half ? w[(3 + rr) * 3 + 2] : w[rr * 3 + 2];
                    Uc[rr] = w0 * M[0];
                    Uc[LD + rr] = w0 * M[1] + w1 * M[3];
                    Uc[2 * LD + rr] = w0 * M[2] + w1 * M[4] + w2 * M[5];
                }
                if (slot == 0 && half == 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++) im[(pl * 3 + c) * LD + rows_used] = pb[6 + c];
                }
            }
            lds_barrier();
        }
    } else {
    // ---- consumers ----------------------------------------------------------------------------------------------------
    const int cw = wave - 4, ct = tid - 256;
    const int bi0 = cw == 3 ? 0 : cw, bi1 = cw == 3 ? 3 : cw + 1, bi2 = cw < 2 ? cw + 2 : 0;   // the wave's operand blocks
    int t_i0[3], t_j0[3];
    t_i0[0] = (cw == 3 ? 0 : cw) * 16; t_j0[0] = (cw == 3 ? 3 : cw) * 16;
    t_i0[1] = (cw == 3 ? 3 : cw) * 16; t_j0[1] = (cw == 3 ? 3 : cw + 1) * 16;
    t_i0[2] = cw < 2 ? cw * 16 : -1;   t_j0[2] = cw < 2 ? (cw + 2) * 16 : 0;
    const double* x0p = img + (lane / 16) * LD + bi0 * 16 + lane % 16;
    const double* x1p = img + (lane / 16) * LD + bi1 * 16 + lane % 16;
    const double* x2p = img + (lane / 16) * LD + bi2 * 16 + lane % 16;
    schur_double4 acc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) acc[j] = schur_double4{0.0, 0.0, 0.0, 0.0};
    // this lane's pieces of a batch's rows: piece c = ct + 256 s of pair c / 3 (pieces past 3 * PB * K: none)
    int pc_pair[kPiecePasses], pc_part[kPiecePasses];
    bool pc_on[kPiecePasses];
#pragma unroll
    for (int s = 0; s < kPiecePasses; s++) {
        const int c = ct + 256 * s;
        pc_pair[s] = c / 3; pc_part[s] = c - 3 * pc_pair[s];
        pc_on[s] = pc_pair[s] < PB * K;
    }
    const uint32_t* prow = pair_row + (size_t)ch.batch_begin * kS10PairsPerBatch;
    uint32_t rowi[kPiecePasses];
    double2 gpiece[kPiecePasses];
    double2 ptreg = make_double2(0.0, 0.0);
    auto load_index = [&](int rel) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < kPiecePasses; s++) rowi[s] = pc_on[s] ? prow[(size_t)rel * kS10PairsPerBatch + pc_pair[s]] : 0xFFFFFFFFu;
    };
    auto load_batch = [&](int rel) __attribute__((always_inline)) {   // rows through rowi (loaded for `rel`), point block
#pragma unroll
        for (int s = 0; s < kPiecePasses; s++) {
            const bool has = rowi[s] != 0xFFFFFFFFu;
            const double2 v2 = reinterpret_cast<const double2*>(ar.g + kArG * (size_t)(has ? rowi[s] : 0u))[pc_part[s]];
            gpiece[s] = has ? v2 : make_double2(0.0, 0.0);
        }
        if (ct < PB * 6) {
            const SchurBatch bt = batch_at(rel);
            const uint32_t pl6 = (uint32_t)ct / 6, piece = (uint32_t)ct - 6 * pl6;
            const bool live = pl6 < bt.p_end - bt.p_begin;
            const double2 v2 = reinterpret_cast<const double2*>(ptfac + kPtFac * ((size_t)bt.p_begin + (live ? pl6 : 0u)))[piece];
            ptreg = live ? v2 : make_double2(0.0, 0.0);   // past the batch's points: zeros (t must be: the k padding of U~ t)
        }
    };
    auto store_batch = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < kPiecePasses; s++)
            if (pc_on[s]) reinterpret_cast<double2*>(gbuf + par * (kS10PairsPerBatch * 6))[ct + 256 * s] = gpiece[s];
        if (ct < PB * 6) reinterpret_cast<double2*>(ptbuf + par * (PB * kPtFac))[ct] = ptreg;
    };
    load_index(0);
    for (int b = -2; b < nb; b++) {
        const bool more = b + 2 < nb;
        if (more) {
            load_batch(b + 2);
            if (b + 3 < nb) load_index(b + 3);
        }
        if (b >= 0) {
            const int off = (b & 1) * (KB * LD);
            schur_mfma_shared4<KB, LD>(cw, x0p + off, x1p + off, x2p + off, acc);
        }
        if (more) store_batch(b & 1);
        lds_barrier();
    }
    // the chunk's window: [pair a <= b][6x6] then [camera][6] (layout and symmetry rule as in ba_schur_kernel); column 6 K of
    // the tiles holds U~ t.  Assembled in LDS (the images are done with) so that it leaves as one contiguous stream.
    double* out = img;
    const int n_pair = K * (K + 1) / 2;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (t_i0[j] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = t_i0[j] + 4 * r + lane / 16, col = t_j0[j] + lane % 16;
            if (col == rows_used && row < rows_used) out[n_pair * 36 + row] = acc[j][r];
            if (row >= rows_used || col >= rows_used) continue;
            const int a = row / 6, rr = row - 6 * a, b = col / 6, cc = col - 6 * b;
            if (a > b) continue;
            const int pidx = a * K - a * (a - 1) / 2 + (b - a);
            double* blk = out + pidx * 36;
            const double v = acc[j][r];
            if (a < b) {
                blk[rr * 6 + cc] = v;
            } else if (row <= col) {
                blk[rr * 6 + cc] = v;
                if (row < col) blk[cc * 6 + rr] = v;
            }
        }
    }
    }
    lds_barrier();
    {
        double* dst = slab + chunk_slab[blockIdx.x];
        const int n_out = K * (K + 1) / 2 * 36 + rows_used;
        for (int i = tid; i < n_out; i += NT) dst[i] = img[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// Long tracks: points outside the windowed kernel's limits (more than 32 free cameras or more than kBatchObs
// observations - a landmark watched while the vehicle stands still).  Same arithmetic, no window: W and Y of
// every observation go to a scratch array, then one 36-lane group per camera pair (host-built list, fixed order)
// writes Y_a W_b^T to that pair's own slab slot.  No atomics; ba_schur_reduce adds the slots like any chunk's.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ba_long_prepare_kernel(const LongPoint* __restrict__ long_pts, const uint32_t* __restrict__ lo_row,
                                                             const uint32_t* __restrict__ lo_cam, const uint32_t* __restrict__ lo_cam_off,
                                                             const double* __restrict__ ar, const double* __restrict__ campre,
                                                             const double* __restrict__ pts, const double* __restrict__ C,
                                                             const double* __restrict__ gp, const double* __restrict__ sp,
                                                             const LmDiag lm, double* __restrict__ Cinv, double* __restrict__ wy,
                                                             double* __restrict__ slab, double* __restrict__ scal)
{
    const LongPoint lp = long_pts[blockIdx.x];
    const size_t p = lp.p;
    double m[6], ci[6];
#pragma unroll
    for (int i = 0; i < 6; i++) m[i] = C[6 * p + i];
    m[0] += point_lambda(m[0], sp[3 * p], lm);
    m[3] += point_lambda(m[3], sp[3 * p + 1], lm);
    m[5] += point_lambda(m[5], sp[3 * p + 2], lm);
    const bool ok = sym3_inverse(m, ci);
    if (threadIdx.x == 0) {
        if (!ok) scal[SC_SCHUR_STATUS] = 1.0;
#pragma unroll
        for (int i = 0; i < 6; i++) Cinv[6 * p + i] = ci[i];
    }
    const double g0 = gp[3 * p], g1 = gp[3 * p + 1], g2 = gp[3 * p + 2];
    const double x[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
    for (uint32_t lo = lp.lo_begin + threadIdx.x; lo < lp.lo_end; lo += 64) {
        const double* a = ar + kArG * (size_t)lo_row[lo];
        const double* pc = campre + kPoseStride * (size_t)lo_cam[lo];
        double G[6], R[9], v[3], w[18];
#pragma unroll
        for (int i = 0; i < 6; i++) G[i] = a[i];
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = pc[i];
        compact_v(R, pc[21] != 0.0, x, v);
        compact_wt(G, R, v, w);
        double* out = wy + 36 * (size_t)lo;
        double* cam = slab + lo_cam_off[lo];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double w0 = w[r * 3], w1 = w[r * 3 + 1], w2 = w[r * 3 + 2];
            const double y0 = w0 * ci[0] + w1 * ci[1] + w2 * ci[2];
            const double y1 = w0 * ci[1] + w1 * ci[3] + w2 * ci[4];
            const double y2 = w0 * ci[2] + w1 * ci[4] + w2 * ci[5];
            out[r * 3] = w0; out[r * 3 + 1] = w1; out[r * 3 + 2] = w2;
            out[18 + r * 3] = y0; out[18 + r * 3 + 1] = y1; out[18 + r * 3 + 2] = y2;
            cam[r] = y0 * g0 + y1 * g1 + y2 * g2;
        }
    }
}

__global__ __launch_bounds__(256) void ba_long_pairs_kernel(uint32_t n_pairs, const uint32_t* __restrict__ pair_a,
                                                            const uint32_t* __restrict__ pair_b, const uint32_t* __restrict__ pair_off,
                                                            const double* __restrict__ wy, double* __restrict__ slab)
{
    const int t = threadIdx.x % 36;
    const uint32_t pair = blockIdx.x * 7 + threadIdx.x / 36;
    if (threadIdx.x >= 252 || pair >= n_pairs) return;
    const double* Y = wy + 36 * (size_t)pair_a[pair] + 18 + (t / 6) * 3;
    const double* W = wy + 36 * (size_t)pair_b[pair] + (t % 6) * 3;
    slab[pair_off[pair] + t] = Y[0] * W[0] + Y[1] * W[1] + Y[2] * W[2];
}

// S = B (diagonal blocks) - T_a (sum of the chunk windows) T_b^T, rhs = -g_c + T_a (sum of the chunk rhs parts), in the
// fixed order of the host-built contribution lists; T = blockdiag(M^T, I) is the per-camera factor the point kernels
// leave out (ba_device.h).  Also exports diag(B) and g_c for the all-reduce.  Overwrites S and rhs completely.
__global__ __launch_bounds__(64) void ba_schur_reduce_kernel(uint32_t n_blocks, uint32_t n_free, const uint32_t* __restrict__ blk_ptr,
                                                             const uint32_t* __restrict__ blk_off, const uint32_t* __restrict__ cam_ptr,
                                                             const uint32_t* __restrict__ cam_off, const uint32_t* __restrict__ blk_row,
                                                             const uint32_t* __restrict__ blk_col, const uint32_t* __restrict__ free_cam,
                                                             const double* __restrict__ campre, const double* __restrict__ slab,
                                                             const uint32_t* __restrict__ cam_tile_start,
                                                             const double* __restrict__ tile_part,
                                                             double* __restrict__ S, double* __restrict__ rhs,
                                                             double* __restrict__ diagB, double* __restrict__ gc_red,
                                                             const double* __restrict__ cost_in, double* __restrict__ cost_out,
                                                             const CamDamp damp, const int with_damp)
{
    __shared__ double t0[36], t1[36];
    __shared__ double bt[36], gt[6], Tm[36], Bc[36], gc6[6];   // the camera's own block from the linearisation's tile sums
    const uint32_t id = blockIdx.x;
    const int t = threadIdx.x;
    if (id == 0 && t == 0) *cost_out = *cost_in;   // this rank's cost joins the reduce payload (saves a 4-us copy command)
    if (id < n_blocks) {
        const uint32_t fa = blk_row[id], fb = blk_col[id];
        const int r = t / 6, c = t % 6;
        if (fa == fb) cam_block_from_tiles(free_cam[fa], cam_tile_start, tile_part, campre, bt, gt, Tm, Bc, gc6);   // block-uniform
        {
            // the contributions in list order; their offsets arrive 64 at a time, one per lane, and reach the adding lanes as lane
            // reads - the slab loads of a group are independent of each other and of any index load (a small window's block
            // collects one contribution per chunk: ~200 at the reference's 20-frame windows)
            double s = 0.0;
            const uint32_t e1 = blk_ptr[id + 1];
            for (uint32_t e0 = blk_ptr[id]; e0 < e1; e0 += 64) {
                const uint32_t n = min(64u, e1 - e0);
                const uint32_t off = (uint32_t)t < n ? blk_off[e0 + t] : 0u;
                uint32_t j = 0;
                for (; j + 8 <= n; j += 8) {   // eight loads in flight, added in list order
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(j + u));
                        v[u] = t < 36 ? slab[o + t] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) s += v[u];
                }
                for (; j < n; j++) {
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)j);
                    if (t < 36) s += slab[o + t];
                }
            }
            if (t < 36) t0[t] = s;
        }
        __syncthreads();
        if (t < 36) {
            // rows: U = T_a S~, U[r][c] = sum_i M_a[i][r] S~[i][c] for r < 3
            const double* Ma = campre + kPoseStride * (size_t)free_cam[fa] + 9;
            t1[t] = r < 3 ? Ma[r] * t0[c] + Ma[3 + r] * t0[6 + c] + Ma[6 + r] * t0[12 + c] : t0[t];
        }
        __syncthreads();
        double v = 0.0;
        if (t < 36) {
            // columns: (U T_b^T)[r][c] = sum_j U[r][j] M_b[j][c] for c < 3
            const double* Mb = campre + kPoseStride * (size_t)free_cam[fb] + 9;
            v = c < 3 ? t1[r * 6] * Mb[c] + t1[r * 6 + 1] * Mb[3 + c] + t1[r * 6 + 2] * Mb[6 + c] : t1[t];
        }
        __syncthreads();
        if (t < 36) t0[t] = v;
        __syncthreads();
        if (t < 36) {
            // a same-camera block is symmetric only to rounding after the two products: its lower entries take the
            // upper ones, so every consumer that mirrors or reads either triangle sees the same bits
            const double w = (fa == fb && r > c) ? t0[c * 6 + r] : v;
            double out = (fa == fb ? Bc[t] : 0.0) - w;
            if (with_damp && fa == fb && r == c) {
                // the camera damping (what ba_cam_damp does in a launch of its own after an all-reduce): one rank, no band factor
                const uint32_t i = 6 * fa + (uint32_t)r;
                const double d = Bc[t];
                double sc = damp.sc[i];
                if (damp.init_scale) {
                    sc = damp.jacobi ? 1.0 / (1.0 + sqrt(d)) : 1.0;
                    damp.sc[i] = sc;
                }
                const double lam = point_lambda(d, sc, damp.lm);
                damp.lc[i] = lam;
                out += lam;
            }
            S[36 * (size_t)id + t] = out;
        }
    } else {
        const uint32_t f = id - n_blocks;
        if (f >= n_free) return;
        cam_block_from_tiles(free_cam[f], cam_tile_start, tile_part, campre, bt, gt, Tm, Bc, gc6);
        {
            double s = 0.0;
            const uint32_t e1 = cam_ptr[f + 1];
            for (uint32_t e0 = cam_ptr[f]; e0 < e1; e0 += 64) {
                const uint32_t n = min(64u, e1 - e0);
                const uint32_t off = (uint32_t)t < n ? cam_off[e0 + t] : 0u;
                uint32_t j = 0;
                for (; j + 8 <= n; j += 8) {   // eight loads in flight, added in list order
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(j + u));
                        v[u] = t < 6 ? slab[o + t] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) s += v[u];
                }
                for (; j < n; j++) {
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)j);
                    if (t < 6) s += slab[o + t];
                }
            }
            if (t < 6) t0[t] = s;
        }
        __syncthreads();
        if (t < 6) {
            const double* M = campre + kPoseStride * (size_t)free_cam[f] + 9;
            const double v = t < 3 ? M[t] * t0[0] + M[3 + t] * t0[1] + M[6 + t] * t0[2] : t0[t];
            const double g = gc6[t];
            rhs[6 * (size_t)f + t] = v - g;
            diagB[6 * (size_t)f + t] = Bc[t * 7];
            gc_red[6 * (size_t)f + t] = g;
        }
    }
}

// after the all-reduce: Jacobi scale (first linearisation only), camera damping onto the diagonal of S
__global__ __launch_bounds__(256) void ba_cam_damp_kernel(uint32_t n_free, const double* __restrict__ diagB,
                                                          double* __restrict__ sc, int init_scale, int jacobi,
                                                          const LmDiag lm, const int32_t* __restrict__ diag_block,
                                                          double* __restrict__ S, double* __restrict__ lc)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_free * 6) return;
    const uint32_t f = i / 6, a = i % 6;
    const double d = diagB[i];
    double s = sc[i];
    if (init_scale) {
        s = jacobi ? 1.0 / (1.0 + sqrt(d)) : 1.0;
        sc[i] = s;
    }
    const double lam = point_lambda(d, s, lm);
    lc[i] = lam;
    S[36 * (size_t)diag_block[f] + a * 7] += lam;
}

// candidate cameras, full-length camera step, per-workgroup partials of the camera share of the step scalars, and the
// candidate's pose table (what a ba_pose_prepare launch of its own did): a workgroup holds kCamUpdateCams whole cameras
__global__ __launch_bounds__(256) void ba_cam_update_kernel(uint32_t n_cam, const int32_t* __restrict__ cam_free,
                                                            const double* __restrict__ cams,
                                                            const double* __restrict__ dc_free,
                                                            const double* __restrict__ lc, const double* __restrict__ gc_red,
                                                            const double* __restrict__ lin_resid,
                                                            const double* __restrict__ campre,
                                                            double* __restrict__ cams_out, double* __restrict__ dc_full,
                                                            double* __restrict__ dcw, double* __restrict__ cam_part,
                                                            double* __restrict__ campre_c)
{
    __shared__ double red[4 * 5];
    __shared__ double xcand[kCamUpdateCams * 6];
    double mcc = 0.0, st2 = 0.0, x2 = 0.0, gd = 0.0, gm = 0.0;
    const uint32_t c = blockIdx.x * kCamUpdateCams + threadIdx.x / 6, a = threadIdx.x % 6;
    const uint32_t i = c * 6 + a;
    if (threadIdx.x < kCamUpdateCams * 6 && c < n_cam) {
        const int32_t f = cam_free[c];
        const double x = cams[i];
        double d = 0.0;
        if (f >= 0) {
            const uint32_t fi = 6 * (uint32_t)f + a;
            d = dc_free[fi];
            const double g = gc_red[fi];
            mcc += 0.5 * (lc[fi] * d * d - g * d);
            if (lin_resid) mcc += 0.5 * d * lin_resid[fi];
            gd += g * d;
            x2 += x * x;
            gm = fmax(gm, fabs(g));
        }
        const double xn = x + d;
        const double e = xn - x;
        st2 += e * e;
        cams_out[i] = xn;
        xcand[threadIdx.x] = xn;
        dc_full[i] = d;
        // what ba_backsub needs per camera: w = M dc_rot (the rotation part of J_c dc is A (w x v)) and dc_t; read from
        // the solver's output directly, so no lane depends on another lane's store
        double wv = d;
        if (a < 3) {
            wv = 0.0;
            if (f >= 0) {
                const double* M = campre + kPoseStride * (size_t)c + 9 + a * 3;
                const double* dr = dc_free + 6 * (size_t)f;
                wv = M[0] * dr[0] + M[1] * dr[1] + M[2] * dr[2];
            }
        }
        dcw[i] = wv;
    }
    mcc = wave_sum(mcc); st2 = wave_sum(st2); x2 = wave_sum(x2); gd = wave_sum(gd); gm = wave_max(gm);
    if (threadIdx.x % kWave == 0) {
        double* o = red + (threadIdx.x / kWave) * 5;
        o[0] = mcc; o[1] = st2; o[2] = x2; o[3] = gd; o[4] = gm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0, d = 0, e = 0;
        for (int w = 0; w < 4; w++) { a += red[w * 5]; b += red[w * 5 + 1]; c += red[w * 5 + 2]; d += red[w * 5 + 3]; e = fmax(e, red[w * 5 + 4]); }
        double* o = cam_part + 5 * (size_t)blockIdx.x;   // summed over the workgroups by ba_step_sums
        o[0] = a; o[1] = b; o[2] = c; o[3] = d; o[4] = e;
    }
    // the candidate's rotation block, one lane per camera of this workgroup (xcand was stored before the barrier above)
    if (campre_c && threadIdx.x < kCamUpdateCams) {
        const uint32_t cc = blockIdx.x * kCamUpdateCams + threadIdx.x;
        if (cc < n_cam) pose_prepare(xcand + 6 * threadIdx.x, campre_c + kPoseStride * (size_t)cc);
    }
}

// K8  back-substitution dp = -Cinv (g_p + sum_obs J_p^T (J_c dc)), candidate point with the box bounds,
// per-workgroup partials {model-cost share, |step|^2, |x|^2, g.step, max|g|}.
// From the compact rows: J_c dc = A u with u = w x v + dc_t (w = M dc_rot per camera, from ba_cam_update; v = R x or x),
// J_p^T (A u) = R^T (G u).
__global__ __launch_bounds__(kPointBlock) void ba_backsub_kernel(
    uint32_t n_pt, const uint32_t* __restrict__ pt_start, const uint32_t* __restrict__ pt_obs,
    const uint32_t* __restrict__ q_cam, const double* __restrict__ ar, const double* __restrict__ campre,
    const double* __restrict__ dcw, const double* __restrict__ Cinv, const double* __restrict__ C,
    const double* __restrict__ gp, const double* __restrict__ sp, const double* __restrict__ pts, const LmDiag lm,
    const double bound_lo, const double bound_hi, double* __restrict__ pts_out, double* __restrict__ dp_out,
    double* __restrict__ part)
{
    __shared__ double red[(kPointBlock / kWave) * 5];
    // kBacksubLanes lanes per point share its observations (more rows in flight per CU: the kernel is a gather),
    // their partial sums meet in a fixed butterfly; lane 0 of the group finishes the point
    const uint32_t p = (blockIdx.x * kPointBlock + threadIdx.x) / kBacksubLanes;
    const uint32_t sub = threadIdx.x % kBacksubLanes;
    double mcc = 0.0, st2 = 0.0, x2 = 0.0, gd = 0.0, gm = 0.0;
    const bool live = p < n_pt;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (live) {
        const double x[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
#pragma unroll 2
        for (uint32_t q = pt_start[p] + sub; q < pt_start[p + 1]; q += kBacksubLanes) {
            const size_t k = pt_obs[q];
            const size_t cam = q_cam[q];
            const double2* row = reinterpret_cast<const double2*>(ar + kArG * k);
            const double* pc = campre + kPoseStride * cam;
            const double* d = dcw + 6 * cam;
            double G[6], R[10];
#pragma unroll
            for (int i = 0; i < 3; i++) { const double2 v2 = row[i]; G[2 * i] = v2.x; G[2 * i + 1] = v2.y; }
#pragma unroll
            for (int i = 0; i < 5; i++) { const double2 v2 = reinterpret_cast<const double2*>(pc)[i]; R[2 * i] = v2.x; R[2 * i + 1] = v2.y; }
            const bool small = pc[21] != 0.0;
            const double w0 = d[0], w1 = d[1], w2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
            double v[3];
            compact_v(R, small, x, v);
            const double u0 = w1 * v[2] - w2 * v[1] + d3, u1 = w2 * v[0] - w0 * v[2] + d4, u2 = w0 * v[1] - w1 * v[0] + d5;
            const double z0 = G[0] * u0 + G[1] * u1 + G[2] * u2;     // G u = A^T (A u)
            const double z1 = G[1] * u0 + G[3] * u1 + G[4] * u2;
            const double z2 = G[2] * u0 + G[4] * u1 + G[5] * u2;
            t0 += R[0] * z0 + R[3] * z1 + R[6] * z2;
            t1 += R[1] * z0 + R[4] * z1 + R[7] * z2;
            t2 += R[2] * z0 + R[5] * z1 + R[8] * z2;
        }
    }
    t0 = group_sum<kBacksubLanes>(t0); t1 = group_sum<kBacksubLanes>(t1); t2 = group_sum<kBacksubLanes>(t2);
    if (live && sub == 0) {
        const double g[3] = {gp[3 * (size_t)p], gp[3 * (size_t)p + 1], gp[3 * (size_t)p + 2]};
        t0 += g[0]; t1 += g[1]; t2 += g[2];
        const double* ci = Cinv + 6 * (size_t)p;
        const double e[3] = {-(ci[0] * t0 + ci[1] * t1 + ci[2] * t2), -(ci[1] * t0 + ci[3] * t1 + ci[4] * t2),
                             -(ci[2] * t0 + ci[4] * t1 + ci[5] * t2)};
        const double* Cp = C + 6 * (size_t)p;
        const double cd[3] = {Cp[0], Cp[3], Cp[5]};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double x = pts[3 * (size_t)p + a];
            const double xn = fmin(fmax(x + e[a], bound_lo), bound_hi);
            const double st = xn - x;
            pts_out[3 * (size_t)p + a] = xn;
            dp_out[3 * (size_t)p + a] = e[a];
            mcc += 0.5 * (point_lambda(cd[a], sp[3 * (size_t)p + a], lm) * e[a] * e[a] - g[a] * e[a]);
            st2 += st * st;
            x2 += x * x;
            gd += g[a] * e[a];
            gm = fmax(gm, fabs(g[a]));
        }
    }
    mcc = wave_sum(mcc); st2 = wave_sum(st2); x2 = wave_sum(x2); gd = wave_sum(gd); gm = wave_max(gm);
    if (threadIdx.x % kWave == 0) {
        double* o = red + (threadIdx.x / kWave) * 5;
        o[0] = mcc; o[1] = st2; o[2] = x2; o[3] = gd; o[4] = gm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0, d = 0, e = 0;
        for (int w = 0; w < kPointBlock / kWave; w++) { a += red[w * 5]; b += red[w * 5 + 1]; c += red[w * 5 + 2]; d += red[w * 5 + 3]; e = fmax(e, red[w * 5 + 4]); }
        double* o = part + 5 * (size_t)blockIdx.x;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d; o[4] = e;
    }
}


// ---- coarse basis of the two-level PCG (pcg_multi.hip: pcg2_solve) for reduced camera systems no band factor applies to --------
// The weak directions of S are the drift modes of the chain: whole stretches of cameras (with their points) moved rigidly cost
// next to nothing.  P_c (6 x 6, row-major; rows [dw (3) | dt (3)], columns [v (3) | theta (3)]) is camera c's parameter change
// under a rigid motion x -> x + theta x (x - a) + v of its aggregate's share of the world (a = the centre of the aggregate's
// reference camera):  R' = R (I - [theta]x)  =>  dw = -M^-1 R theta  (d(R x)/dw = -[R x]x M, M of the pose table),
// dt = -R v - R [a]x theta.
__global__ __launch_bounds__(256) void ba_coarse_basis_kernel(const uint32_t n_free, const uint32_t* __restrict__ free_cam,
                                                              const uint32_t* __restrict__ row_agg, const uint32_t* __restrict__ agg_ref,
                                                              const double* __restrict__ campre, double* __restrict__ P)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_free) return;
    const double* pc = campre + kPoseStride * (size_t)free_cam[f];
    const double* pr = campre + kPoseStride * (size_t)agg_ref[row_agg[f]];
    double R[9], M[9], a[3];
#pragma unroll
    for (int i = 0; i < 9; i++) { R[i] = pc[i]; M[i] = pc[9 + i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) a[i] = -(pr[i] * pr[18] + pr[3 + i] * pr[19] + pr[6 + i] * pr[20]);   // -R_ref^T t_ref
    // M^-1 by the adjugate (M = R Jr: well conditioned away from |w| = 2 pi)
    double Mi[9];
    {
        const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
        const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
        const double id = 1.0 / det;
        Mi[0] = c00 * id; Mi[1] = (M[2] * M[7] - M[1] * M[8]) * id; Mi[2] = (M[1] * M[5] - M[2] * M[4]) * id;
        Mi[3] = c01 * id; Mi[4] = (M[0] * M[8] - M[2] * M[6]) * id; Mi[5] = (M[2] * M[3] - M[0] * M[5]) * id;
        Mi[6] = c02 * id; Mi[7] = (M[1] * M[6] - M[0] * M[7]) * id; Mi[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    }
    const double ax[9] = {0.0, -a[2], a[1], a[2], 0.0, -a[0], -a[1], a[0], 0.0};
    double* Pc = P + 36 * (size_t)f;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double mr = 0.0, ra = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) { mr += Mi[i * 3 + k] * R[k * 3 + j]; ra += R[i * 3 + k] * ax[k * 3 + j]; }
            Pc[i * 6 + j] = 0.0;
            Pc[i * 6 + 3 + j] = -mr;
            Pc[(3 + i) * 6 + j] = -R[i * 3 + j];
            Pc[(3 + i) * 6 + 3 + j] = -ra;
        }
}

}  // namespace

// ---- launch wrappers --------------------------------------------------------------------------------

void launch_ba_coarse_basis(hipStream_t s, uint32_t n_free, const uint32_t* free_cam, const uint32_t* row_agg, const uint32_t* agg_ref,
                            const double* campre, double* P)
{
    if (!n_free) return;
    hipLaunchKernelGGL(ba_coarse_basis_kernel, dim3((n_free + 255) / 256), dim3(256), 0, s, n_free, free_cam, row_agg, agg_ref, campre, P);
}

void launch_pose_prepare(hipStream_t s, uint32_t n_cam, const double* cams, double* campre)
{
    if (!n_cam) return;
    hipLaunchKernelGGL(ba_pose_prepare_kernel, dim3((n_cam + 63) / 64), dim3(64), 0, s, n_cam, cams, campre);
}

void launch_linearize(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt,
                      const double* campre, const double* pts, const int32_t* cam_free, const Proj& P, double delta,
                      CompactRows ar, double* tile_part, const double* gate)
{
    if (!n_tiles) return;
    hipLaunchKernelGGL(ba_linearize_kernel, dim3(n_tiles), dim3(kTileThreads), 0, s, tiles, uv, obs_pt, campre, pts,
                       cam_free, P, delta, ar, tile_part, gate);
}

void launch_debug_rows(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt,
                       const double* campre, const double* pts, const int32_t* cam_free, const Proj& P, double delta,
                       double* r_out, double* jc_out, double* jp_out)
{
    if (!n_tiles) return;
    hipLaunchKernelGGL(ba_debug_rows_kernel, dim3(n_tiles), dim3(kTileThreads), 0, s, tiles, uv, obs_pt, campre, pts, cam_free, P, delta,
                       r_out, jc_out, jp_out);
}

void launch_cost(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt,
                 const double* campre, const double* pts, const Proj& P, double delta, double* cost_part)
{
    if (!n_tiles) return;
    hipLaunchKernelGGL(ba_cost_kernel, dim3(n_tiles), dim3(kTileThreads), 0, s, tiles, uv, obs_pt, campre, pts, P, delta, cost_part);
}

void launch_sum_strided(hipStream_t s, const double* in, uint32_t n, uint32_t stride, uint32_t offset, double scale, double* out)
{
    hipLaunchKernelGGL(sum_strided_kernel, dim3(1), dim3(1024), 0, s, in, n, stride, offset, scale, out, Publish{nullptr, nullptr, nullptr, 0ull, 0, 0, 0});
}

// see PackedSeg: block (i, j) moves slice j of segment i in 16-byte units, slice 0 the odd bytes at the end
constexpr int kPackSlices = 8;
__global__ __launch_bounds__(256) void packed_scatter_kernel(const PackedSeg* __restrict__ table, const unsigned char* __restrict__ payload)
{
    const PackedSeg sg = table[blockIdx.x];
    const bool fill = sg.src_offset == ~0ull;
    // (payload == nullptr: src_offset is the source's address itself - device arrays gathered into pinned host memory)
    const unsigned char* const sbytes = payload ? payload + (fill ? 0 : sg.src_offset) : reinterpret_cast<const unsigned char*>(fill ? 0 : sg.src_offset);
    const uint4* src = reinterpret_cast<const uint4*>(sbytes);
    uint4* dst = reinterpret_cast<uint4*>(sg.dst);
    const uint64_t units = sg.bytes / 16;
    for (uint64_t u = blockIdx.y * 256u + threadIdx.x; u < units; u += kPackSlices * 256u) dst[u] = fill ? uint4{0u, 0u, 0u, 0u} : src[u];
    if (blockIdx.y == 0 && threadIdx.x < (sg.bytes & 15)) {
        const uint64_t b = units * 16 + threadIdx.x;
        reinterpret_cast<unsigned char*>(sg.dst)[b] = fill ? (unsigned char)0 : sbytes[b];
    }
}

void launch_packed_scatter(hipStream_t s, const PackedSeg* table, int n_seg, const unsigned char* payload)
{
    if (n_seg > 0) hipLaunchKernelGGL(packed_scatter_kernel, dim3((unsigned)n_seg, kPackSlices), dim3(256), 0, s, table, payload);
}

// The scalars the host decides on, written straight into pinned host memory, then a sequence number: the host
// polls that word instead of waiting for a copy command and its completion signal.
// The scalars the host decides on go straight into pinned host memory, followed by a sequence number: the host polls
// that word instead of waiting for a copy command and its completion signal.  Normally the tail of the iteration's
// last kernel (the candidate-cost sum); a kernel of its own when a collective comes after that sum.
__global__ __launch_bounds__(64) void ba_publish_kernel(const Publish pb) { publish_tail(pb); }

void launch_publish(hipStream_t s, double* src, int n, int clear_first, int clear_n, double* host_dst, unsigned long long* host_seq,
                    unsigned long long seq)
{
    hipLaunchKernelGGL(ba_publish_kernel, dim3(1), dim3(64), 0, s, Publish{src, host_dst, host_seq, seq, n, clear_first, clear_n});
}

void launch_step_sums(hipStream_t s, const double* part5, uint32_t n5, double* out5, const double* cam5, uint32_t n_cam5, double* out_cam5,
                      const double* cost_part, uint32_t n_cost, double* out_cost, double* gate, const double* status, double x_cost,
                      double min_relative_decrease, int gate_enabled, double stop_vote, double* pub_src, int n_pub, int clear_first,
                      int clear_n, double* host_dst, unsigned long long* host_seq, unsigned long long seq, int armijo_in_gate)
{
    hipLaunchKernelGGL(ba_step_sums_kernel, dim3(1), dim3(1024), 0, s,
                       StepSumsArgs{part5, n5, out5, cam5, n_cam5, out_cam5, cost_part, n_cost, out_cost,
                                    StepGate{gate, status, x_cost, min_relative_decrease, gate_enabled, armijo_in_gate}, stop_vote,
                                    Publish{pub_src, host_dst, host_seq, seq, n_pub, clear_first, clear_n}});
}

void launch_gate_publish(hipStream_t s, double* scal, double x_cost, double min_relative_decrease, int gate_enabled, double* pub_src,
                         int n_pub, int clear_first, int clear_n, double* host_dst, unsigned long long* host_seq, unsigned long long seq,
                         int armijo_in_gate)
{
    hipLaunchKernelGGL(ba_gate_publish_kernel, dim3(1), dim3(64), 0, s, scal, x_cost, min_relative_decrease, gate_enabled, armijo_in_gate,
                       Publish{pub_src, host_dst, host_seq, seq, n_pub, clear_first, clear_n});
}

void launch_sum5(hipStream_t s, const double* in, uint32_t n, double* out)
{
    hipLaunchKernelGGL(sum5_kernel, dim3(1), dim3(1024), 0, s, in, n, out);
}

void launch_point_reduce(hipStream_t s, uint32_t n_pt, const uint32_t* pt_start, const uint32_t* pt_obs, const uint32_t* q_cam,
                         CompactRows ar, const double* campre, double* C, double* gp, const double* gate)
{
    if (!n_pt) return;
    hipLaunchKernelGGL(ba_point_reduce_kernel, dim3(backsub_blocks(n_pt)), dim3(kPointBlock), 0, s, n_pt, pt_start, pt_obs, q_cam, ar,
                       campre, C, gp, gate);
}

void launch_point_scale(hipStream_t s, uint32_t n_pt, const double* C, int jacobi, double* sp)
{
    if (!n_pt) return;
    hipLaunchKernelGGL(ba_point_scale_kernel, dim3((n_pt + kPointBlock - 1) / kPointBlock), dim3(kPointBlock), 0, s, n_pt, C,
                       jacobi, sp);
}

void launch_schur(hipStream_t s, int kmax, uint32_t n_chunks, const SchurChunk* chunks, const SchurBatch* batches,
                  const uint32_t* chunk_slab, const uint32_t* chunk_cam, const uint32_t* pair_row, const uint32_t* pt_obs, const uint32_t* q_pt,
                  const uint8_t* q_slot, CompactRows ar, const double* campre, const double* pts, double* C, double* gp,
                  const double* sp, LmDiag lm, double* Cinv, double* ptfac, double* slab, double* scal, const uint32_t* pt_start,
                  const uint32_t* q_cam, int point_blocks_from_rows)
{
    if (!n_chunks) return;
    if (kmax <= 10) {
        // windows of at most 10 cameras: the pipelined kernel (kS10Lds = 59 KB of LDS, two workgroups of eight waves per CU)
        static bool attr_set[64] = {};   // once per DEVICE (the attribute belongs to the function on a device): the call is not free and sits right behind the host's decision
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev = dev >= 0 && dev < 64 ? dev : 0;
        if (!attr_set[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur10_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS10Lds);
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(ba_schur10_kernel, dim3(n_chunks), dim3(kS10Threads), kS10Lds, s, chunks, batches, chunk_slab, chunk_cam, pair_row, ar,
                           campre, pts, C, gp, sp, lm, Cinv, ptfac, slab, scal, pt_start, pt_obs, q_cam, q_slot, point_blocks_from_rows);
    } else if (kmax <= 16) {
        constexpr size_t lds = SchurShape<16>::lds_bytes;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ba_schur_kernel<16>), dim3(n_chunks), dim3(schur_threads(16)), lds, s, chunks, batches, chunk_slab, chunk_cam, pt_obs,
                           q_pt, q_slot, ar.g, campre, pts, C, gp, sp, lm, Cinv, ptfac, slab, scal);
    } else if (kmax <= 20) {
        // the reference's 20-frame sliding windows: 120 window rows = eight tile rows, 36 tiles instead of the 78 of a 32-camera image
        constexpr size_t lds = SchurShape<20>::lds_bytes;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_kernel<20>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ba_schur_kernel<20>), dim3(n_chunks), dim3(schur_threads(20)), lds, s, chunks, batches, chunk_slab, chunk_cam, pt_obs,
                           q_pt, q_slot, ar.g, campre, pts, C, gp, sp, lm, Cinv, ptfac, slab, scal);
    } else {
        constexpr size_t lds = SchurShape<32>::lds_bytes;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ba_schur_kernel<32>), dim3(n_chunks), dim3(schur_threads(32)), lds, s, chunks, batches, chunk_slab, chunk_cam, pt_obs,
                           q_pt, q_slot, ar.g, campre, pts, C, gp, sp, lm, Cinv, ptfac, slab, scal);
    }
}

// ---- line search trial (TrustRegionMinimizer::DoLineSearch on bounded problems; run only when the full step fails the
// sufficient-decrease condition) ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_ls_candidate_kernel(uint32_t n_cam, uint32_t n_pt, const double* __restrict__ cams,
                                                              const double* __restrict__ pts, const double* __restrict__ dc_full,
                                                              const double* __restrict__ dp, const double a, const double lo, const double hi,
                                                              double* __restrict__ cams_out, double* __restrict__ pts_out,
                                                              double* __restrict__ ls_part, double* __restrict__ campre_c)
{
    __shared__ double red[8];
    double st2 = 0.0, dmax = 0.0;
    const uint32_t nb_pt = (n_pt * 3 + 255) / 256;
    if (blockIdx.x < nb_pt) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i < n_pt * 3) {
            const double x = pts[i], d = dp[i];
            const double xn = fmin(fmax(x + a * d, lo), hi);   // ParameterBlock::Plus projects onto the bounds
            pts_out[i] = xn;
            st2 = (xn - x) * (xn - x);
            dmax = fabs(d);
        }
    } else {
        for (uint32_t i = threadIdx.x; i < n_cam * 6; i += 256) {
            const double d = dc_full[i];
            cams_out[i] = fma(a, d, cams[i]);   // (spelled out: the pose table below must be that of exactly these values)
            dmax = fmax(dmax, fabs(d));
        }
        // the trial's pose table (what a ba_pose_prepare launch did): a lane per camera, from the same sums
        for (uint32_t c = threadIdx.x; c < n_cam; c += 256) {
            double cc[6];
#pragma unroll
            for (int k = 0; k < 6; k++) cc[k] = fma(a, dc_full[6 * (size_t)c + k], cams[6 * (size_t)c + k]);
            pose_prepare(cc, campre_c + kPoseStride * (size_t)c);
        }
    }
    st2 = wave_sum(st2); dmax = wave_max(dmax);
    if (threadIdx.x % kWave == 0) { red[threadIdx.x / kWave] = st2; red[4 + threadIdx.x / kWave] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ls_part[2 * (size_t)blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        ls_part[2 * (size_t)blockIdx.x + 1] = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    }
}

__global__ __launch_bounds__(kTileThreads) void ba_ls_eval_kernel(
    const Tile* __restrict__ tiles, const float4* __restrict__ uv, const uint32_t* __restrict__ obs_pt,
    const double* __restrict__ campre_c, const double* __restrict__ pts_c, const double* __restrict__ dc_full,
    const double* __restrict__ dp, const int32_t* __restrict__ cam_free, const Proj P, const double delta,
    double* __restrict__ tile_part2)
{
    __shared__ double red[2 * (kTileThreads / kWave)];
    const Tile t = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    PosePre pr;
    pose_load(campre_c + kPoseStride * (size_t)t.cam, pr);
    const bool fixed = cam_free[t.cam] < 0;
    // direction of the camera block in camera-frame terms: J_c dc = A (D dc_rot + dc_t), D = -[v]x M  =>  D dc_rot = (M dc_rot) x v
    const double* dcam = dc_full + 6 * (size_t)t.cam;
    double w[3] = {0.0, 0.0, 0.0}, dt[3] = {0.0, 0.0, 0.0};
    if (!fixed) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            w[i] = pr.M[i * 3] * dcam[0] + pr.M[i * 3 + 1] * dcam[1] + pr.M[i * 3 + 2] * dcam[2];
            dt[i] = dcam[3 + i];
        }
    }
    double rho = 0.0, dir = 0.0;
    for (uint32_t o = tid; o < t.count; o += kTileThreads) {
        const size_t k = (size_t)t.start + o;
        const uint32_t p = obs_pt[k];
        const double x[3] = {pts_c[3 * (size_t)p], pts_c[3 * (size_t)p + 1], pts_c[3 * (size_t)p + 2]};
        const double e[3] = {dp[3 * (size_t)p], dp[3 * (size_t)p + 1], dp[3 * (size_t)p + 2]};
        double r[4], am[12], D[9];
        rho += residual_ad(pr, x, uv[k], P, delta, r, am, D);
        double v[3];
        compact_v(pr.R, pr.small, x, v);
        double u[3] = {w[1] * v[2] - w[2] * v[1] + dt[0], w[2] * v[0] - w[0] * v[2] + dt[1], w[0] * v[1] - w[1] * v[0] + dt[2]};
#pragma unroll
        for (int i = 0; i < 3; i++) u[i] += pr.R[i * 3] * e[0] + pr.R[i * 3 + 1] * e[1] + pr.R[i * 3 + 2] * e[2];   // J_p dp = A R dp
#pragma unroll
        for (int i = 0; i < 4; i++) dir += r[i] * (am[i * 3] * u[0] + am[i * 3 + 1] * u[1] + am[i * 3 + 2] * u[2]);
    }
    rho = wave_sum(rho); dir = wave_sum(dir);
    if (tid % kWave == 0) { red[2 * (tid / kWave)] = rho; red[2 * (tid / kWave) + 1] = dir; }
    __syncthreads();
    if (tid < 2) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < kTileThreads / kWave; q++) s += red[2 * q + tid];
        tile_part2[2 * (size_t)blockIdx.x + tid] = s;
    }
}

__global__ __launch_bounds__(1024) void ba_ls_sums_kernel(const double* __restrict__ tile_part2, uint32_t n_tiles,
                                                           const double* __restrict__ ls_part, uint32_t n_ls_part,
                                                           double* __restrict__ scal, const Publish pb)
{
    __shared__ double red[16 * 4];
    double c = 0.0, d = 0.0, s2 = 0.0, mx = 0.0;
    for (uint32_t i = threadIdx.x; i < n_tiles; i += 1024) { c += tile_part2[2 * (size_t)i]; d += tile_part2[2 * (size_t)i + 1]; }
    for (uint32_t i = threadIdx.x; i < n_ls_part; i += 1024) { s2 += ls_part[2 * (size_t)i]; mx = fmax(mx, ls_part[2 * (size_t)i + 1]); }
    c = wave_sum(c); d = wave_sum(d); s2 = wave_sum(s2); mx = wave_max(mx);
    if (threadIdx.x % kWave == 0) { double* o = red + 4 * (threadIdx.x / kWave); o[0] = c; o[1] = d; o[2] = s2; o[3] = mx; }
    __syncthreads();
    static_assert(SC_LS_DIR == SC_LS_COST + 1 && SC_LS_STEP2 == SC_LS_COST + 2 && SC_LS_DMAX == SC_LS_COST + 3, "line-search slots");
    if (threadIdx.x < 4) {
        double a = 0.0;
        for (int q = 0; q < 16; q++) a = threadIdx.x == 3 ? fmax(a, red[4 * q + 3]) : a + red[4 * q + threadIdx.x];
        scal[SC_LS_COST + threadIdx.x] = threadIdx.x == 0 ? 0.5 * a : a;
    }
    publish_tail(pb);
}

void launch_ls_candidate(hipStream_t s, uint32_t n_cam, uint32_t n_pt, const double* cams, const double* pts, const double* dc_full,
                         const double* dp, double a, double bound_lo, double bound_hi, double* cams_out, double* pts_out, double* ls_part,
                         double* campre_c)
{
    hipLaunchKernelGGL(ba_ls_candidate_kernel, dim3(ls_candidate_blocks(n_pt)), dim3(256), 0, s, n_cam, n_pt, cams, pts, dc_full, dp, a,
                       bound_lo, bound_hi, cams_out, pts_out, ls_part, campre_c);
}

void launch_ls_eval(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt, const double* campre_c,
                    const double* pts_c, const double* dc_full, const double* dp, const int32_t* cam_free, const Proj& P, double delta,
                    double* tile_part2)
{
    if (!n_tiles) return;
    hipLaunchKernelGGL(ba_ls_eval_kernel, dim3(n_tiles), dim3(kTileThreads), 0, s, tiles, uv, obs_pt, campre_c, pts_c, dc_full, dp, cam_free, P,
                       delta, tile_part2);
}

void launch_ls_sums(hipStream_t s, const double* tile_part2, uint32_t n_tiles, const double* ls_part, uint32_t n_ls_part, double* scal,
                    double* pub_src, int n_pub, double* host_dst, unsigned long long* host_seq, unsigned long long seq)
{
    hipLaunchKernelGGL(ba_ls_sums_kernel, dim3(1), dim3(1024), 0, s, tile_part2, n_tiles, ls_part, n_ls_part, scal,
                       Publish{pub_src, host_dst, host_seq, seq, n_pub, 0, 0});
}

// Staging of the host-collective leg: device <-> pinned host memory by a kernel, in stream order with everything else
// (the same mechanism as the publication of the step scalars; no copy engine, no second queue).
__global__ __launch_bounds__(256) void ba_copy_f64_kernel(double* __restrict__ dst, const double* __restrict__ src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
}

void launch_copy_f64(hipStream_t s, double* dst, const double* src, uint64_t n)
{
    if (!n) return;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(ba_copy_f64_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, dst, src, n);
}

// a sequence number into pinned host memory once everything before it on the stream has completed and is visible to the host
__global__ void ba_flag_kernel(unsigned long long* __restrict__ host_seq, const unsigned long long seq)
{
    __threadfence_system();
    __atomic_store_n(host_seq, seq, __ATOMIC_RELEASE);
}

void launch_flag(hipStream_t s, unsigned long long* host_seq, unsigned long long seq)
{
    hipLaunchKernelGGL(ba_flag_kernel, dim3(1), dim3(1), 0, s, host_seq, seq);
}

__global__ void ba_status_poison_kernel(double* __restrict__ scal)
{
    if (scal[SC_SCHUR_STATUS] != 0.0) scal[SC_CAND_COST] = __builtin_huge_val();
}

void launch_status_poison(hipStream_t s, double* scal) { hipLaunchKernelGGL(ba_status_poison_kernel, dim3(1), dim3(1), 0, s, scal); }

void launch_schur_long(hipStream_t s, uint32_t n_long, const LongPoint* long_pts, const uint32_t* lo_row, const uint32_t* lo_cam,
                       const uint32_t* lo_cam_off, uint32_t n_pairs, const uint32_t* pair_a, const uint32_t* pair_b,
                       const uint32_t* pair_off, const double* ar, const double* campre, const double* pts, const double* C,
                       const double* gp, const double* sp, LmDiag lm, double* Cinv, double* wy, double* slab, double* scal)
{
    if (!n_long) return;
    hipLaunchKernelGGL(ba_long_prepare_kernel, dim3(n_long), dim3(64), 0, s, long_pts, lo_row, lo_cam, lo_cam_off, ar, campre, pts, C, gp,
                       sp, lm, Cinv, wy, slab, scal);
    if (n_pairs)
        hipLaunchKernelGGL(ba_long_pairs_kernel, dim3((n_pairs + 6) / 7), dim3(256), 0, s, n_pairs, pair_a, pair_b, pair_off, wy, slab);
}

void launch_schur_reduce(hipStream_t s, uint32_t n_blocks, uint32_t n_free, const uint32_t* blk_ptr, const uint32_t* blk_off,
                         const uint32_t* cam_ptr, const uint32_t* cam_off, const uint32_t* blk_row, const uint32_t* blk_col,
                         const uint32_t* free_cam, const double* campre, const double* slab, const uint32_t* cam_tile_start,
                         const double* tile_part, double* S, double* rhs, double* diagB, double* gc_red, const double* cost_in,
                         double* cost_out, const CamDamp* damp)
{
    if (!(n_blocks + n_free)) {
        (void)hipMemcpyAsync(cost_out, cost_in, sizeof(double), hipMemcpyDeviceToDevice, s);
        return;
    }
    hipLaunchKernelGGL(ba_schur_reduce_kernel, dim3(n_blocks + n_free), dim3(64), 0, s, n_blocks, n_free, blk_ptr, blk_off, cam_ptr,
                       cam_off, blk_row, blk_col, free_cam, campre, slab, cam_tile_start, tile_part, S, rhs, diagB, gc_red, cost_in, cost_out,
                       damp ? *damp : CamDamp{}, damp ? 1 : 0);
}

void launch_cam_damp(hipStream_t s, uint32_t n_free, const double* diagB, double* sc, int init_scale, int jacobi,
                     LmDiag lm, const int32_t* diag_block, double* S, double* lc)
{
    if (!n_free) return;
    hipLaunchKernelGGL(ba_cam_damp_kernel, dim3((n_free * 6 + 255) / 256), dim3(256), 0, s, n_free, diagB, sc, init_scale, jacobi,
                       lm, diag_block, S, lc);
}

void launch_cam_update(hipStream_t s, uint32_t n_cam, const int32_t* cam_free, const double* cams,
                       const double* dc_free, const double* lc, const double* gc_red, const double* lin_resid, const double* campre,
                       double* cams_out, double* dc_full, double* dcw, double* cam_part, double* campre_c)
{
    hipLaunchKernelGGL(ba_cam_update_kernel, dim3(cam_update_blocks(n_cam)), dim3(256), 0, s, n_cam, cam_free, cams, dc_free, lc, gc_red,
                       lin_resid, campre, cams_out, dc_full, dcw, cam_part, campre_c);
}

// ---- small problems (the reference's windows: up to 32 cameras): candidate cameras, back-substitution and candidate cost in one launch --
// ba_cam_update, ba_backsub and ba_cost are 5 + 6 + 5 us of launch-sized kernels at a 20-frame window, each waiting for the one
// before.  With few cameras every workgroup can form all candidate cameras and their pose table itself (LDS), so the three
// become one: per point the step (as ba_backsub), the candidate, and - point-major, through the pt_obs index - the candidate's
// residuals.  Workgroup 0 also writes what the others only use: candidate cameras, dc_full, w, the pose table, and the camera
// share of the step scalars.  cost_part gets one entry per workgroup (ba_step_sums adds them as it adds ba_cost's tiles).
constexpr int kApplySmallCams = 32;
__global__ __launch_bounds__(kPointBlock) void ba_apply_small_kernel(
    uint32_t n_cam, const int32_t* __restrict__ cam_free, const double* __restrict__ cams, const double* __restrict__ dc_free,
    const double* __restrict__ lc, const double* __restrict__ gc_red, const double* __restrict__ lin_resid,
    double* __restrict__ cams_out, double* __restrict__ dc_full, double* __restrict__ dcw_out, double* __restrict__ cam_part,
    double* __restrict__ campre_c,
    uint32_t n_pt, const uint32_t* __restrict__ pt_start, const uint32_t* __restrict__ pt_obs, const uint32_t* __restrict__ q_cam,
    const double* __restrict__ ar, const double* __restrict__ campre, const double* __restrict__ Cinv, const double* __restrict__ C,
    const double* __restrict__ gp, const double* __restrict__ sp, const double* __restrict__ pts, const LmDiag lm,
    const double bound_lo, const double bound_hi, double* __restrict__ pts_out, double* __restrict__ dp_out, double* __restrict__ part,
    const float4* __restrict__ uv, const Proj P, const double delta, double* __restrict__ cost_part,
    unsigned int* __restrict__ arrivals, const StepSumsArgs sums, const int with_sums)
{
    __shared__ double red[(kPointBlock / kWave) * 11];
    __shared__ double fin[11];
    __shared__ int s_last;
    // with_sums: the workgroup that finishes last also does what ba_step_sums does in a launch of its own.  Every partial then
    // leaves as a write-through (sc1) store of lane 0, which drains them and adds to the arrival counter (MI355X_MICROARCH.md,
    // "Valid forms", row 1); the last workgroup reads them with sc1 loads
    auto put = [&](double* ptr, double val) __attribute__((always_inline)) {
        if (with_sums) __hip_atomic_store(ptr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *ptr = val;
    };
    __shared__ double s_x[kApplySmallCams * 6], s_w[kApplySmallCams * 6];
    __shared__ double s_pre[kApplySmallCams * kPoseStride];
    const int tid = threadIdx.x;
    const bool first = blockIdx.x == 0;
    // ---- the cameras (every workgroup; workgroup 0 publishes) ----
    {
        double mcc = 0.0, st2 = 0.0, x2 = 0.0, gd = 0.0, gm = 0.0;
        const uint32_t c = (uint32_t)tid / 6, a = (uint32_t)tid % 6;
        if (tid < (int)(n_cam * 6)) {
            const int32_t f = cam_free[c];
            const double x = cams[tid];
            double d = 0.0;
            if (f >= 0) {
                const uint32_t fi = 6 * (uint32_t)f + a;
                d = dc_free[fi];
                const double g = gc_red[fi];
                mcc += 0.5 * (lc[fi] * d * d - g * d);
                if (lin_resid) mcc += 0.5 * d * lin_resid[fi];
                gd += g * d;
                x2 += x * x;
                gm = fmax(gm, fabs(g));
            }
            const double xn = x + d;
            const double e = xn - x;
            st2 += e * e;
            s_x[tid] = xn;
            double wv = d;   // w = M dc_rot (rotation part of J_c dc is A (w x v)) | dc_t
            if (a < 3) {
                wv = 0.0;
                if (f >= 0) {
                    const double* M = campre + kPoseStride * (size_t)c + 9 + a * 3;
                    const double* dr = dc_free + 6 * (size_t)f;
                    wv = M[0] * dr[0] + M[1] * dr[1] + M[2] * dr[2];
                }
            }
            s_w[tid] = wv;
            if (first) { cams_out[tid] = xn; dc_full[tid] = d; dcw_out[tid] = wv; }
        }
        if (first) {
            mcc = wave_sum(mcc); st2 = wave_sum(st2); x2 = wave_sum(x2); gd = wave_sum(gd); gm = wave_max(gm);
            if (tid % kWave == 0) {
                double* o = red + (tid / kWave) * 6;
                o[0] = mcc; o[1] = st2; o[2] = x2; o[3] = gd; o[4] = gm;
            }
        }
        __syncthreads();
        if (first && tid == 0) {
            double a0 = 0, b0 = 0, c0 = 0, d0 = 0, e0 = 0;
            for (int w = 0; w < kPointBlock / kWave; w++) { a0 += red[w * 6]; b0 += red[w * 6 + 1]; c0 += red[w * 6 + 2]; d0 += red[w * 6 + 3]; e0 = fmax(e0, red[w * 6 + 4]); }
            put(cam_part, a0); put(cam_part + 1, b0); put(cam_part + 2, c0); put(cam_part + 3, d0); put(cam_part + 4, e0);
        }
        if (tid < (int)n_cam) {
            pose_prepare(s_x + 6 * tid, s_pre + kPoseStride * tid);
            if (first) {
#pragma unroll
                for (int i = 0; i < kPoseStride; i++) campre_c[kPoseStride * (size_t)tid + i] = s_pre[kPoseStride * tid + i];
            }
        }
        __syncthreads();
    }
    // ---- the points: step (ba_backsub), candidate, the candidate's cost ----
    const uint32_t p = (blockIdx.x * kPointBlock + (uint32_t)tid) / kBacksubLanes;
    const uint32_t sub = (uint32_t)tid % kBacksubLanes;
    double mcc = 0.0, st2 = 0.0, x2 = 0.0, gd = 0.0, gm = 0.0, rho = 0.0;
    const bool live = p < n_pt;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    uint32_t q0 = 0, q1 = 0;
    double x[3] = {0.0, 0.0, 0.0};
    if (live) {
        x[0] = pts[3 * (size_t)p]; x[1] = pts[3 * (size_t)p + 1]; x[2] = pts[3 * (size_t)p + 2];
        q0 = pt_start[p]; q1 = pt_start[p + 1];
#pragma unroll 2
        for (uint32_t q = q0 + sub; q < q1; q += kBacksubLanes) {
            const size_t k = pt_obs[q];
            const size_t cam = q_cam[q];
            const double2* row = reinterpret_cast<const double2*>(ar + kArG * k);
            const double* pc = campre + kPoseStride * cam;
            const double* d = s_w + 6 * cam;
            double G[6], R[10];
#pragma unroll
            for (int i = 0; i < 3; i++) { const double2 v2 = row[i]; G[2 * i] = v2.x; G[2 * i + 1] = v2.y; }
#pragma unroll
            for (int i = 0; i < 5; i++) { const double2 v2 = reinterpret_cast<const double2*>(pc)[i]; R[2 * i] = v2.x; R[2 * i + 1] = v2.y; }
            const bool small = pc[21] != 0.0;
            const double w0 = d[0], w1 = d[1], w2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
            double v[3];
            compact_v(R, small, x, v);
            const double u0 = w1 * v[2] - w2 * v[1] + d3, u1 = w2 * v[0] - w0 * v[2] + d4, u2 = w0 * v[1] - w1 * v[0] + d5;
            const double z0 = G[0] * u0 + G[1] * u1 + G[2] * u2;     // G u = A^T (A u)
            const double z1 = G[1] * u0 + G[3] * u1 + G[4] * u2;
            const double z2 = G[2] * u0 + G[4] * u1 + G[5] * u2;
            t0 += R[0] * z0 + R[3] * z1 + R[6] * z2;
            t1 += R[1] * z0 + R[4] * z1 + R[7] * z2;
            t2 += R[2] * z0 + R[5] * z1 + R[8] * z2;
        }
    }
    t0 = group_sum<kBacksubLanes>(t0); t1 = group_sum<kBacksubLanes>(t1); t2 = group_sum<kBacksubLanes>(t2);   // (in every lane of the group)
    double xn[3] = {0.0, 0.0, 0.0};
    if (live) {
        const double g[3] = {gp[3 * (size_t)p], gp[3 * (size_t)p + 1], gp[3 * (size_t)p + 2]};
        t0 += g[0]; t1 += g[1]; t2 += g[2];
        const double* ci = Cinv + 6 * (size_t)p;
        const double e[3] = {-(ci[0] * t0 + ci[1] * t1 + ci[2] * t2), -(ci[1] * t0 + ci[3] * t1 + ci[4] * t2),
                             -(ci[2] * t0 + ci[4] * t1 + ci[5] * t2)};
#pragma unroll
        for (int a = 0; a < 3; a++) xn[a] = fmin(fmax(x[a] + e[a], bound_lo), bound_hi);
        if (sub == 0) {
            const double* Cp = C + 6 * (size_t)p;
            const double cd[3] = {Cp[0], Cp[3], Cp[5]};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double st = xn[a] - x[a];
                pts_out[3 * (size_t)p + a] = xn[a];
                dp_out[3 * (size_t)p + a] = e[a];
                mcc += 0.5 * (point_lambda(cd[a], sp[3 * (size_t)p + a], lm) * e[a] * e[a] - g[a] * e[a]);
                st2 += st * st;
                x2 += x[a] * x[a];
                gd += g[a] * e[a];
                gm = fmax(gm, fabs(g[a]));
            }
        }
        // the candidate's residuals, the group's lanes sharing the observations again; pose table of the candidate cameras in LDS
        for (uint32_t q = q0 + sub; q < q1; q += kBacksubLanes) {
            PosePre pr;
            pose_load(s_pre + kPoseStride * (size_t)q_cam[q], pr);
            rho += residual_cost(pr, xn, uv[pt_obs[q]], P, delta);
        }
    }
    mcc = wave_sum(mcc); st2 = wave_sum(st2); x2 = wave_sum(x2); gd = wave_sum(gd); gm = wave_max(gm); rho = wave_sum(rho);
    if (tid % kWave == 0) {
        double* o = red + (tid / kWave) * 6;
        o[0] = mcc; o[1] = st2; o[2] = x2; o[3] = gd; o[4] = gm; o[5] = rho;
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0, b = 0, c = 0, d = 0, e = 0, r = 0;
        for (int w = 0; w < kPointBlock / kWave; w++) {
            a += red[w * 6]; b += red[w * 6 + 1]; c += red[w * 6 + 2]; d += red[w * 6 + 3]; e = fmax(e, red[w * 6 + 4]); r += red[w * 6 + 5];
        }
        double* o = part + 5 * (size_t)blockIdx.x;
        put(o, a); put(o + 1, b); put(o + 2, c); put(o + 3, d); put(o + 4, e);
        put(cost_part + blockIdx.x, r);
    }
    if (!with_sums) return;
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int before = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = before == gridDim.x - 1 ? 1 : 0;
        if (s_last) __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    step_sums_body<kPointBlock, true>(sums, red, fin);
}

// (every workgroup forms all candidate cameras for itself, and every workgroup adds to ONE arrival counter: a shape for small
// problems - up to 64 k points, 1 024 workgroups)
bool apply_small_fits(uint32_t n_cam, uint32_t n_pt)
{
    return n_cam >= 1 && n_cam <= (uint32_t)kApplySmallCams && n_pt >= 1 && backsub_blocks(n_pt) <= 1024u;
}

void launch_apply_small(hipStream_t s, uint32_t n_cam, const int32_t* cam_free, const double* cams, const double* dc_free, const double* lc,
                        const double* gc_red, const double* lin_resid, double* cams_out, double* dc_full, double* dcw, double* cam_part,
                        double* campre_c, uint32_t n_pt, const uint32_t* pt_start, const uint32_t* pt_obs, const uint32_t* q_cam, const double* ar,
                        const double* campre, const double* Cinv, const double* C, const double* gp, const double* sp, const double* pts, LmDiag lm,
                        double bound_lo, double bound_hi, double* pts_out, double* dp, double* part, const float4* uv, const Proj& P, double delta,
                        double* cost_part, unsigned int* arrivals, const StepSumsLaunch* sums)
{
    StepSumsArgs q{};
    if (sums)
        q = StepSumsArgs{part, backsub_blocks(n_pt), sums->out5, cam_part, 1u, sums->out_cam5, cost_part, backsub_blocks(n_pt), sums->out_cost,
                         StepGate{sums->gate, sums->status, sums->x_cost, sums->min_relative_decrease, sums->gate_enabled, sums->armijo_in_gate},
                         sums->stop_vote, Publish{sums->pub_src, sums->host_dst, sums->host_seq, sums->seq, sums->n_pub, sums->clear_first, sums->clear_n}};
    hipLaunchKernelGGL(ba_apply_small_kernel, dim3(backsub_blocks(n_pt)), dim3(kPointBlock), 0, s, n_cam, cam_free, cams, dc_free, lc, gc_red,
                       lin_resid, cams_out, dc_full, dcw, cam_part, campre_c, n_pt, pt_start, pt_obs, q_cam, ar, campre, Cinv, C, gp, sp, pts, lm,
                       bound_lo, bound_hi, pts_out, dp, part, uv, P, delta, cost_part, arrivals, q, sums ? 1 : 0);
}

void launch_backsub(hipStream_t s, uint32_t n_pt, const uint32_t* pt_start, const uint32_t* pt_obs, const uint32_t* q_cam,
                    const double* ar, const double* campre, const double* dcw,
                    const double* Cinv, const double* C, const double* gp, const double* sp, const double* pts,
                    LmDiag lm, double bound_lo, double bound_hi, double* pts_out, double* dp, double* part)
{
    if (!n_pt) return;
    hipLaunchKernelGGL(ba_backsub_kernel, dim3(backsub_blocks(n_pt)), dim3(kPointBlock), 0, s, n_pt, pt_start,
                       pt_obs, q_cam, ar, campre, dcw, Cinv, C, gp, sp, pts, lm, bound_lo, bound_hi, pts_out, dp, part);
}

}  // namespace soslam
