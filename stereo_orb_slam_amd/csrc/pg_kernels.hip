// pg_kernels.hip - pose-graph kernels for gfx950: SE(3) edge linearisation with block scatter, robust chi2,
// vertex update.  Semantics of g2o's VertexSE3 / EdgeSE3 / RobustKernelHuber as the reference configures them
// (/root/reference/src/pose_graph_optimizer.cpp:6-27,98-173).
//
// BASELINE.json configs[4] is 5k vertices / 20k edges: ~12 MB touched per iteration, so every kernel here is
// launch/latency-bound, not HBM-bound; one lane per edge or vertex, f64 atomics for the
// block scatter (2.4 M adds per linearisation, far below the atomic rate).
#include "pg_kernels.h"
#include "reduce.h"

namespace soslam {

namespace {

struct Quat { double x, y, z, w; };

__device__ __forceinline__ Quat qmul(const Quat a, const Quat b)
{
    return Quat{a.w * b.x + b.w * a.x + a.y * b.z - a.z * b.y, a.w * b.y + b.w * a.y + a.z * b.x - a.x * b.z,
                a.w * b.z + b.w * a.z + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
__device__ __forceinline__ Quat qconj(const Quat a) { return Quat{-a.x, -a.y, -a.z, a.w}; }
__device__ __forceinline__ Quat qnormalized(Quat q)
{
    const double n = 1.0 / sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x *= n; q.y *= n; q.z *= n; q.w *= n;
    return q;
}
__device__ __forceinline__ void qrot(const Quat q, double* __restrict__ R)
{
    R[0] = 1 - 2 * (q.y * q.y + q.z * q.z); R[1] = 2 * (q.x * q.y - q.z * q.w);     R[2] = 2 * (q.x * q.z + q.y * q.w);
    R[3] = 2 * (q.x * q.y + q.z * q.w);     R[4] = 1 - 2 * (q.x * q.x + q.z * q.z); R[5] = 2 * (q.y * q.z - q.x * q.w);
    R[6] = 2 * (q.x * q.z - q.y * q.w);     R[7] = 2 * (q.y * q.z + q.x * q.w);     R[8] = 1 - 2 * (q.x * q.x + q.y * q.y);
}

// e = toVectorMQT(Z^-1 Xi^-1 Xj); optionally the two 6x6 Jacobians wrt the [dt, dq.xyz] increments
template <bool WITH_J>
__device__ __forceinline__ void edge_error(const double* __restrict__ xi, const double* __restrict__ xj,
                                           const double* __restrict__ z, double* __restrict__ e,
                                           double* __restrict__ ji, double* __restrict__ jj)
{
    const Quat qi = qnormalized(Quat{xi[3], xi[4], xi[5], xi[6]});
    const Quat qj = qnormalized(Quat{xj[3], xj[4], xj[5], xj[6]});
    const Quat qz = qnormalized(Quat{z[3], z[4], z[5], z[6]});
    const Quat qa = qmul(qconj(qi), qj);
    Quat qe = qmul(qconj(qz), qa);
    double Ri[9], Rz[9];
    qrot(qi, Ri);
    qrot(qz, Rz);
    const double d[3] = {xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2]};
    double ta[3], te[3];
#pragma unroll
    for (int i = 0; i < 3; i++) ta[i] = Ri[i] * d[0] + Ri[3 + i] * d[1] + Ri[6 + i] * d[2];
    const double u[3] = {ta[0] - z[0], ta[1] - z[1], ta[2] - z[2]};
#pragma unroll
    for (int i = 0; i < 3; i++) te[i] = Rz[i] * u[0] + Rz[3 + i] * u[1] + Rz[6 + i] * u[2];
    const double sgn = qe.w < 0.0 ? -1.0 : 1.0;
    qe.x *= sgn; qe.y *= sgn; qe.z *= sgn; qe.w *= sgn;
    e[0] = te[0]; e[1] = te[1]; e[2] = te[2]; e[3] = qe.x; e[4] = qe.y; e[5] = qe.z;
    if (WITH_J) {
        double Re[9];
        qrot(qe, Re);
#pragma unroll
        for (int i = 0; i < 36; i++) { ji[i] = 0.0; jj[i] = 0.0; }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                jj[r * 6 + c] = Re[r * 3 + c];        // d te / d dt_j
                ji[r * 6 + c] = -Rz[c * 3 + r];       // d te / d dt_i = -Rz^T
            }
        // d qe / d dq_j = w I + [v]x
        jj[21] = qe.w;  jj[22] = -qe.z; jj[23] = qe.y;
        jj[27] = qe.z;  jj[28] = qe.w;  jj[29] = -qe.x;
        jj[33] = -qe.y; jj[34] = qe.x;  jj[35] = qe.w;
        // d te / d dq_i = 2 Rz^T [ta]x
        const double tx[9] = {0, -ta[2], ta[1], ta[2], 0, -ta[0], -ta[1], ta[0], 0};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                ji[r * 6 + 3 + c] = 2.0 * (Rz[r] * tx[c] + Rz[3 + r] * tx[3 + c] + Rz[6 + r] * tx[6 + c]);
        // d qe / d dq_i : column k = -sgn * vec(conj(qz) * e_k * qa)
        const Quat qzc = qconj(qz);
        const Quat ex = qmul(qmul(qzc, Quat{1, 0, 0, 0}), qa), ey = qmul(qmul(qzc, Quat{0, 1, 0, 0}), qa), ez = qmul(qmul(qzc, Quat{0, 0, 1, 0}), qa);
        ji[21] = -sgn * ex.x; ji[22] = -sgn * ey.x; ji[23] = -sgn * ez.x;
        ji[27] = -sgn * ex.y; ji[28] = -sgn * ey.y; ji[29] = -sgn * ez.y;
        ji[33] = -sgn * ex.z; ji[34] = -sgn * ey.z; ji[35] = -sgn * ez.z;
    }
}

__device__ __forceinline__ double quad6(const double* __restrict__ info, const double* __restrict__ e)
{
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < 6; b++) t += info[a * 6 + b] * e[b];
        s += e[a] * t;
    }
    return s;
}

__device__ __forceinline__ void huber_chi(double e2, double delta, double& rho0, double& rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
    else { const double s = sqrt(e2); rho0 = 2.0 * s * delta - dsqr; rho1 = delta / s; }
}

// out = A^T (W B) for 6x6 row-major blocks
__device__ __forceinline__ void atwb(const double* __restrict__ A, const double* __restrict__ WB, double* __restrict__ out)
{
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 6; m++) s += A[m * 6 + a] * WB[m * 6 + c];
            out[a * 6 + c] = s;
        }
}

__global__ __launch_bounds__(128) void pg_linearize_kernel(uint32_t n_edge, const double* __restrict__ est,
                                                           const uint32_t* __restrict__ ef, const uint32_t* __restrict__ et,
                                                           const double* __restrict__ meas, const PgInfo info, const double delta,
                                                           const PgEdgeBlocks* __restrict__ eb, double* __restrict__ econ,
                                                           double* __restrict__ chi_part,
                                                           double* __restrict__ dbg_e, double* __restrict__ dbg_ji,
                                                           double* __restrict__ dbg_jj)
{
    __shared__ double red[2];
    const uint32_t k = blockIdx.x * 128 + threadIdx.x;
    double rho0 = 0.0;
    if (k < n_edge) {
        double e[6], ji[36], jj[36];
        edge_error<true>(est + 7 * (size_t)ef[k], est + 7 * (size_t)et[k], meas + 7 * (size_t)k, e, ji, jj);
        if (dbg_e) {
#pragma unroll
            for (int i = 0; i < 6; i++) dbg_e[6 * (size_t)k + i] = e[i];
#pragma unroll
            for (int i = 0; i < 36; i++) { dbg_ji[36 * (size_t)k + i] = ji[i]; dbg_jj[36 * (size_t)k + i] = jj[i]; }
        }
        double rho1;
        huber_chi(quad6(info.m, e), delta, rho0, rho1);
        // W = rho' Omega ; We, W Ji, W Jj
        double We[6], WJ[36], blk[36];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) s += info.m[a * 6 + c] * e[c];
            We[a] = rho1 * s;
        }
        const PgEdgeBlocks bl = eb[k];
        // vertex i side
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) s += info.m[a * 6 + m] * ji[m * 6 + c];
                WJ[a * 6 + c] = rho1 * s;
            }
        // every edge writes its own record [H_ii | H_jj | H_off as stored | b_i | b_j]; pg_gather sums the records of a
        // block in the fixed order of host-built lists - no floating-point atomics, bitwise reproducible
        double* rec = econ + kPgEdgeRec * (size_t)k;
        if (bl.fi >= 0) {
            atwb(ji, WJ, blk);
#pragma unroll
            for (int i = 0; i < 36; i++) rec[i] = blk[i];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) s += ji[m * 6 + a] * We[m];
                rec[108 + a] = -s;
            }
        }
        if (bl.off >= 0) {
            // stored upper block is (min, max): H_ji = Jj^T W Ji when fj < fi, else H_ij = (Jj^T W Ji)^T
            atwb(jj, WJ, blk);   // Jj^T W Ji : rows of j, columns of i
            if (bl.off_is_ji) {
#pragma unroll
                for (int i = 0; i < 36; i++) rec[72 + i] = blk[i];
            } else {
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int c = 0; c < 6; c++) rec[72 + c * 6 + a] = blk[a * 6 + c];
            }
        }
        if (bl.fj >= 0) {
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double s = 0.0;
#pragma unroll
                    for (int m = 0; m < 6; m++) s += info.m[a * 6 + m] * jj[m * 6 + c];
                    WJ[a * 6 + c] = rho1 * s;
                }
            atwb(jj, WJ, blk);
#pragma unroll
            for (int i = 0; i < 36; i++) rec[36 + i] = blk[i];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++) s += jj[m * 6 + a] * We[m];
                rec[114 + a] = -s;
            }
        }
    }
    rho0 = wave_sum(rho0);
    if (threadIdx.x % 64 == 0) red[threadIdx.x / 64] = rho0;
    __syncthreads();
    if (threadIdx.x == 0) chi_part[blockIdx.x] = red[0] + red[1];
}

// H and b from the per-edge records: workgroup = one block of the upper block-sparse H.  ptr/ent: per block the records
// that feed it, ent = 2 * edge + side (diagonal blocks: side 0 = the edge's i vertex, 1 = its j vertex; the same list
// gives b of that vertex) or the edge index (off-diagonal blocks).  Fixed order, every element of H and b written.
__global__ __launch_bounds__(64) void pg_gather_kernel(uint32_t n_blocks, const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ ent,
                                                       const uint32_t* __restrict__ blk_row, const uint32_t* __restrict__ blk_col,
                                                       const double* __restrict__ econ, double* __restrict__ H, double* __restrict__ b)
{
    const uint32_t id = blockIdx.x;
    const int t = threadIdx.x;
    if (id >= n_blocks || t >= 42) return;
    const bool diag = blk_row[id] == blk_col[id];
    if (!diag && t >= 36) return;
    double s = 0.0;
    for (uint32_t e = ptr[id]; e < ptr[id + 1]; e++) {
        const uint32_t v = ent[e];
        const double* rec = econ + kPgEdgeRec * (size_t)(diag ? v >> 1 : v);
        s += diag ? (t < 36 ? rec[36 * (v & 1) + t] : rec[108 + 6 * (v & 1) + (t - 36)]) : rec[72 + t];
    }
    if (t < 36) H[36 * (size_t)id + t] = s;
    else b[6 * (size_t)blk_row[id] + (t - 36)] = s;
}

__global__ __launch_bounds__(256) void pg_chi2_kernel(uint32_t n_edge, const double* __restrict__ est,
                                                      const uint32_t* __restrict__ ef, const uint32_t* __restrict__ et,
                                                      const double* __restrict__ meas, const PgInfo info, const double delta,
                                                      double* __restrict__ chi_part)
{
    __shared__ double red[4];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    double rho0 = 0.0;
    if (k < n_edge) {
        double e[6], rho1;
        edge_error<false>(est + 7 * (size_t)ef[k], est + 7 * (size_t)et[k], meas + 7 * (size_t)k, e, nullptr, nullptr);
        huber_chi(quad6(info.m, e), delta, rho0, rho1);
    }
    rho0 = wave_sum(rho0);
    if (threadIdx.x % 64 == 0) red[threadIdx.x / 64] = rho0;
    __syncthreads();
    if (threadIdx.x == 0) chi_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = sum of part[0..n) (fixed order), out[1] = max_i |diag(H)_i| when diag_block != nullptr
__global__ __launch_bounds__(1024) void pg_reduce_kernel(const double* __restrict__ part, uint32_t n, const double* __restrict__ H,
                                                         const int32_t* __restrict__ diag_block, uint32_t n_free,
                                                         double* __restrict__ out)
{
    __shared__ double red[16], redm[16];
    double s = 0.0, m = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) s += part[i];
    if (diag_block)
        for (uint32_t i = threadIdx.x; i < n_free * 6; i += 1024) m = fmax(m, fabs(H[36 * (size_t)diag_block[i / 6] + (i % 6) * 7]));
    s = wave_sum(s);
    m = wave_max(m);
    if (threadIdx.x % 64 == 0) { red[threadIdx.x / 64] = s; redm[threadIdx.x / 64] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 16; w++) { a += red[w]; b = fmax(b, redm[w]); }
        out[0] = a;
        if (diag_block) out[1] = b;
    }
}

// candidate estimates X * fromVectorMQT(x) and the gain-ratio denominator x.(lambda x + b) (per-workgroup partials)
__global__ __launch_bounds__(256) void pg_update_kernel(uint32_t n_vertex, const double* __restrict__ est,
                                                        const int32_t* __restrict__ free_idx, const double* __restrict__ x,
                                                        const double* __restrict__ b, const double lambda,
                                                        double* __restrict__ cand, double* __restrict__ scale_part)
{
    __shared__ double red[4];
    const uint32_t v = blockIdx.x * 256 + threadIdx.x;
    double sc = 0.0;
    if (v < n_vertex) {
        const double* s = est + 7 * (size_t)v;
        double* o = cand + 7 * (size_t)v;
        const int32_t f = free_idx[v];
        if (f < 0) {
#pragma unroll
            for (int i = 0; i < 7; i++) o[i] = s[i];
        } else {
            const double* d = x + 6 * (size_t)f;
            const double* bb = b + 6 * (size_t)f;
            const double d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5];
            sc = d0 * (lambda * d0 + bb[0]) + d1 * (lambda * d1 + bb[1]) + d2 * (lambda * d2 + bb[2]) +
                 d3 * (lambda * d3 + bb[3]) + d4 * (lambda * d4 + bb[4]) + d5 * (lambda * d5 + bb[5]);
            const Quat q = qnormalized(Quat{s[3], s[4], s[5], s[6]});
            const double w2 = 1.0 - (d3 * d3 + d4 * d4 + d5 * d5);
            const Quat dq = w2 < 0.0 ? Quat{0, 0, 0, 1} : Quat{d3, d4, d5, sqrt(w2)};   // fromCompactQuaternion
            double R[9];
            qrot(q, R);
            const Quat qn = qnormalized(qmul(q, dq));
            o[0] = s[0] + R[0] * d0 + R[1] * d1 + R[2] * d2;
            o[1] = s[1] + R[3] * d0 + R[4] * d1 + R[5] * d2;
            o[2] = s[2] + R[6] * d0 + R[7] * d1 + R[8] * d2;
            o[3] = qn.x; o[4] = qn.y; o[5] = qn.z; o[6] = qn.w;
        }
    }
    sc = wave_sum(sc);
    if (threadIdx.x % 64 == 0) red[threadIdx.x / 64] = sc;
    __syncthreads();
    if (threadIdx.x == 0) scale_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}


// ---- coarse space of the two-level preconditioner (pcg_multi.hip: pcg2_solve) ---------------------------------------------------
// P_v (6 x 6, row-major): the increment [dt, dq.xyz] of vertex v (g2o's X <- X * fromVectorMQT(increment), a motion in the
// vertex's own frame) under a small rigid motion (v, w) of its whole aggregate about the aggregate's reference position:
//   world motion x -> x + w x (x - c) + v   =>   dt = R^T (v + w x (t - c)),  dq = R^T w / 2.
__global__ __launch_bounds__(256) void pg_coarse_basis_kernel(const uint32_t n_free, const uint32_t* __restrict__ free_vertex,
                                                              const uint32_t* __restrict__ row_agg, const uint32_t* __restrict__ agg_ref,
                                                              const double* __restrict__ est, double* __restrict__ P)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_free) return;
    const double* x = est + 7 * (size_t)free_vertex[f];
    const double* c = est + 7 * (size_t)agg_ref[row_agg[f]];
    double R[9];
    qrot(qnormalized(Quat{x[3], x[4], x[5], x[6]}), R);
    const double t[3] = {x[0] - c[0], x[1] - c[1], x[2] - c[2]};
    // [t]x
    const double tx[9] = {0.0, -t[2], t[1], t[2], 0.0, -t[0], -t[1], t[0], 0.0};
    double* Pv = P + 36 * (size_t)f;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const double rt = R[b * 3 + a];                                    // R^T[a][b]
            double m = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) m += R[k * 3 + a] * tx[k * 3 + b];     // (R^T [t]x)[a][b]
            Pv[a * 6 + b] = rt;
            Pv[a * 6 + 3 + b] = -m;
            Pv[(3 + a) * 6 + b] = 0.0;
            Pv[(3 + a) * 6 + 3 + b] = 0.5 * rt;
        }
}

}  // namespace

void launch_pg_linearize(hipStream_t s, uint32_t n_edge, const double* est, const uint32_t* ef, const uint32_t* et,
                         const double* meas, const PgInfo& info, double delta, const PgEdgeBlocks* eb, double* econ,
                         double* chi_part, double* dbg_e, double* dbg_ji, double* dbg_jj)
{
    if (!n_edge) return;
    hipLaunchKernelGGL(pg_linearize_kernel, dim3((n_edge + 127) / 128), dim3(128), 0, s, n_edge, est, ef, et, meas, info, delta, eb,
                       econ, chi_part, dbg_e, dbg_ji, dbg_jj);
}

void launch_pg_gather(hipStream_t s, uint32_t n_blocks, const uint32_t* ptr, const uint32_t* ent, const uint32_t* blk_row,
                      const uint32_t* blk_col, const double* econ, double* H, double* b)
{
    if (!n_blocks) return;
    hipLaunchKernelGGL(pg_gather_kernel, dim3(n_blocks), dim3(64), 0, s, n_blocks, ptr, ent, blk_row, blk_col, econ, H, b);
}

void launch_pg_chi2(hipStream_t s, uint32_t n_edge, const double* est, const uint32_t* ef, const uint32_t* et,
                    const double* meas, const PgInfo& info, double delta, double* chi_part)
{
    if (!n_edge) return;
    hipLaunchKernelGGL(pg_chi2_kernel, dim3((n_edge + 255) / 256), dim3(256), 0, s, n_edge, est, ef, et, meas, info, delta, chi_part);
}

void launch_pg_reduce(hipStream_t s, const double* part, uint32_t n, const double* H, const int32_t* diag_block, uint32_t n_free,
                      double* out)
{
    hipLaunchKernelGGL(pg_reduce_kernel, dim3(1), dim3(1024), 0, s, part, n, H, diag_block, n_free, out);
}

void launch_pg_update(hipStream_t s, uint32_t n_vertex, const double* est, const int32_t* free_idx, const double* x,
                      const double* b, double lambda, double* cand, double* scale_part)
{
    if (!n_vertex) return;
    hipLaunchKernelGGL(pg_update_kernel, dim3((n_vertex + 255) / 256), dim3(256), 0, s, n_vertex, est, free_idx, x, b, lambda, cand,
                       scale_part);
}

void launch_pg_coarse_basis(hipStream_t s, uint32_t n_free, const uint32_t* free_vertex, const uint32_t* row_agg, const uint32_t* agg_ref,
                            const double* est, double* P)
{
    if (!n_free) return;
    hipLaunchKernelGGL(pg_coarse_basis_kernel, dim3((n_free + 255) / 256), dim3(256), 0, s, n_free, free_vertex, row_agg, agg_ref, est, P);
}

__global__ __launch_bounds__(256) void pg_shift_diag_kernel(double* __restrict__ H, const int32_t* __restrict__ diag_block, uint32_t n_free,
                                                            const double delta)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_free * 6) H[36 * (size_t)diag_block[i / 6] + 7 * (i % 6)] += delta;
}

void launch_pg_shift_diag(hipStream_t s, double* H, const int32_t* diag_block, uint32_t n_free, double delta)
{
    if (n_free) hipLaunchKernelGGL(pg_shift_diag_kernel, dim3((n_free * 6 + 255) / 256), dim3(256), 0, s, H, diag_block, n_free, delta);
}

}  // namespace soslam
