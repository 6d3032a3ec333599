// ba_device.h - per-observation arithmetic of the BA hot path, device side (f64).
//
// Residual:   /root/reference/src/reprojection_error.h:12-41 (rotate by angle-axis, translate,
//             project through two general 3x4 matrices, subtract the stereo measurement).
// Derivative: the reference takes it by autodiff (AutoDiffCostFunction<.,4,6,3>,
//             /root/reference/src/reprojection_error.h:58), i.e. the exact derivative of the branch of
//             ceres::AngleAxisRotatePoint that was taken.  Here it is analytic:
//             d(Rx)/dw = -R [x]x Jr(w) with the closed-form right Jacobian on the Rodrigues branch,
//             -[x]x on the first-order branch.
// Loss:       HuberLoss(delta) with Ceres' corrector (rho'' <= 0 => scale r and J by sqrt(rho')).
#pragma once

#include <hip/hip_runtime.h>

namespace soslam {

struct Proj {
    double l[12];
    double r[12];
};

// rotation matrix of the branch taken by AngleAxisRotatePoint and, if WITH_J, d(R x)/dw
struct PoseRot {
    double R[9];     // row-major; on the small-angle branch this is I + [w]x (not orthonormal - by design)
    double w[3];
    double a, b, c;  // Jr = I - b [w]x + c [w]x^2 ; a unused on the small-angle branch
    bool small;
};

__device__ __forceinline__ void pose_rotation(const double* __restrict__ cam, PoseRot& p)
{
    const double wx = cam[0], wy = cam[1], wz = cam[2];
    p.w[0] = wx; p.w[1] = wy; p.w[2] = wz;
    const double th2 = wx * wx + wy * wy + wz * wz;
    if (th2 > 2.220446049250313e-16) {
        const double th = sqrt(th2);
        // half-angle form: 1 - cos(th) = 2 sin^2(th/2) has no cancellation at small angles
        double sh, ch;
        sincos(0.5 * th, &sh, &ch);
        const double s = 2.0 * sh * ch;
        const double oc = 2.0 * sh * sh;
        const double c = 1.0 - oc;
        const double ith = 1.0 / th;
        const double kx = wx * ith, ky = wy * ith, kz = wz * ith;
        p.R[0] = c + oc * kx * kx;      p.R[1] = -s * kz + oc * kx * ky; p.R[2] = s * ky + oc * kx * kz;
        p.R[3] = s * kz + oc * ky * kx; p.R[4] = c + oc * ky * ky;       p.R[5] = -s * kx + oc * ky * kz;
        p.R[6] = -s * ky + oc * kz * kx; p.R[7] = s * kx + oc * kz * ky; p.R[8] = c + oc * kz * kz;
        // right Jacobian coefficients: Jr = I - (1-cos)/th^2 [w]x + (th - sin)/th^3 [w]x^2
        const double ith2 = ith * ith;
        p.b = oc * ith2;
        // (th - sin th)/th^3: Taylor series below 0.1 rad (next term th^8/39916800 < 3e-16), direct above
        p.c = th2 < 0.01 ? (1.0 / 6.0) - th2 * ((1.0 / 120.0) - th2 * ((1.0 / 5040.0) - th2 * (1.0 / 362880.0)))
                         : (th - s) * ith2 * ith;
        p.small = false;
    } else {
        p.R[0] = 1.0; p.R[1] = -wz; p.R[2] = wy;
        p.R[3] = wz;  p.R[4] = 1.0; p.R[5] = -wx;
        p.R[6] = -wy; p.R[7] = wx;  p.R[8] = 1.0;
        p.b = 0.0; p.c = 0.0;
        p.small = true;
    }
    p.a = 0.0;
}

// Per-camera quantities every observation of that camera needs, computed once per linearisation by
// ba_pose_prepare and read through scalar loads (one camera per workgroup => wave-uniform):
//   R (9)  rotation of the branch taken;  M (9) with d(R x)/dw = -[v]x M, where v = R x and M = R Jr on the Rodrigues
//   branch, v = x and M = I on the first-order branch;  t (3);  small flag.
constexpr int kPoseStride = 24;
struct PosePre {
    double R[9];
    double M[9];
    double t[3];
    bool small;
};

__device__ __forceinline__ void pose_prepare(const double* __restrict__ cam, double* __restrict__ out)
{
    PoseRot p;
    pose_rotation(cam, p);
    const double wx = p.w[0], wy = p.w[1], wz = p.w[2], b = p.b, c = p.c;
    double Jr[9];
    Jr[0] = 1.0 - c * (wy * wy + wz * wz); Jr[1] = b * wz + c * wx * wy;         Jr[2] = -b * wy + c * wx * wz;
    Jr[3] = -b * wz + c * wx * wy;         Jr[4] = 1.0 - c * (wx * wx + wz * wz); Jr[5] = b * wx + c * wy * wz;
    Jr[6] = b * wy + c * wx * wz;          Jr[7] = -b * wx + c * wy * wz;        Jr[8] = 1.0 - c * (wx * wx + wy * wy);
#pragma unroll
    for (int i = 0; i < 9; i++) out[i] = p.R[i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            out[9 + i * 3 + j] = p.small ? (i == j ? 1.0 : 0.0)   // first-order branch: d(x + w x x)/dw = -[x]x, i.e. M = I with v = x
                                         : p.R[i * 3] * Jr[j] + p.R[i * 3 + 1] * Jr[3 + j] + p.R[i * 3 + 2] * Jr[6 + j];
    out[18] = cam[3]; out[19] = cam[4]; out[20] = cam[5];
    out[21] = p.small ? 1.0 : 0.0;
    out[22] = 0.0; out[23] = 0.0;
}

__device__ __forceinline__ void pose_load(const double* __restrict__ pre, PosePre& p)
{
#pragma unroll
    for (int i = 0; i < 9; i++) { p.R[i] = pre[i]; p.M[i] = pre[9 + i]; }
    p.t[0] = pre[18]; p.t[1] = pre[19]; p.t[2] = pre[20];
    p.small = pre[21] != 0.0;
}

// u, v of one camera and the 2x3 derivative wrt the camera-frame point
__device__ __forceinline__ void project_rows(const double* __restrict__ P, const double* p, double& u, double& v,
                                             double* __restrict__ A /* 6 */)
{
    const double d = P[8] * p[0] + P[9] * p[1] + P[10] * p[2] + P[11];
    const double inv = 1.0 / d;
    u = (P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[3]) * inv;
    v = (P[4] * p[0] + P[5] * p[1] + P[6] * p[2] + P[7]) * inv;
    if (A) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            A[i] = (P[i] - u * P[8 + i]) * inv;
            A[3 + i] = (P[4 + i] - v * P[8 + i]) * inv;
        }
    }
}

__device__ __forceinline__ void huber(double s, double delta, double& rho0, double& rho1)
{
    const double b = delta * delta;
    if (s > b) {
        const double r = sqrt(s);
        rho0 = 2.0 * delta * r - b;
        rho1 = fmax(delta / r, 2.2250738585072014e-308);
    } else {
        rho0 = s;
        rho1 = 1.0;
    }
}

// residual only; returns rho(|r|^2)
__device__ __forceinline__ double residual_cost(const PosePre& pr, const double* x, const float4 uv, const Proj& P, double delta)
{
    double y[3];
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = pr.R[i * 3] * x[0] + pr.R[i * 3 + 1] * x[1] + pr.R[i * 3 + 2] * x[2] + pr.t[i];
    double ul, vl, ur, vr;
    project_rows(P.l, y, ul, vl, nullptr);
    project_rows(P.r, y, ur, vr, nullptr);
    const double r0 = ul - (double)uv.x, r1 = vl - (double)uv.y, r2 = ur - (double)uv.z, r3 = vr - (double)uv.w;
    double rho0, rho1;
    huber(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3, delta, rho0, rho1);
    return rho0;
}

// corrected residual r[4], J_c[24] (4x6 row-major), J_p[12] (4x3 row-major); returns rho(|r|^2)
__device__ __forceinline__ double residual_jacobian(const PosePre& pr, const double* x, const float4 uv, const Proj& P,
                                                    double delta, bool cam_fixed, double* __restrict__ r,
                                                    double* __restrict__ jc, double* __restrict__ jp,
                                                    double* __restrict__ a_out = nullptr)
{
    double yr[3], y[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        yr[i] = pr.R[i * 3] * x[0] + pr.R[i * 3 + 1] * x[1] + pr.R[i * 3 + 2] * x[2];
        y[i] = yr[i] + pr.t[i];
    }
    double A[12], ul, vl, ur, vr;
    project_rows(P.l, y, ul, vl, A);
    project_rows(P.r, y, ur, vr, A + 6);
    r[0] = ul - (double)uv.x; r[1] = vl - (double)uv.y; r[2] = ur - (double)uv.z; r[3] = vr - (double)uv.w;
    double rho0, rho1;
    huber(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], delta, rho0, rho1);
    const double sw = sqrt(rho1);

    // D = d(R x)/dw = -[v]x M, 3x3 (v = x, M = I on the first-order branch; v = R x, M = R Jr otherwise)
    double D[9];
    {
        const double v0 = pr.small ? x[0] : yr[0], v1 = pr.small ? x[1] : yr[1], v2 = pr.small ? x[2] : yr[2];
        const double* M = pr.M;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            D[0 * 3 + j] = -(v1 * M[2 * 3 + j] - v2 * M[1 * 3 + j]);
            D[1 * 3 + j] = -(v2 * M[0 * 3 + j] - v0 * M[2 * 3 + j]);
            D[2 * 3 + j] = -(v0 * M[1 * 3 + j] - v1 * M[0 * 3 + j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double a0 = A[i * 3] * sw, a1 = A[i * 3 + 1] * sw, a2 = A[i * 3 + 2] * sw;
        if (a_out) { a_out[i * 3] = a0; a_out[i * 3 + 1] = a1; a_out[i * 3 + 2] = a2; }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            jc[i * 6 + j] = cam_fixed ? 0.0 : a0 * D[j] + a1 * D[3 + j] + a2 * D[6 + j];
            jp[i * 3 + j] = a0 * pr.R[j] + a1 * pr.R[3 + j] + a2 * pr.R[6 + j];
        }
        jc[i * 6 + 3] = cam_fixed ? 0.0 : a0;
        jc[i * 6 + 4] = cam_fixed ? 0.0 : a1;
        jc[i * 6 + 5] = cam_fixed ? 0.0 : a2;
        r[i] *= sw;
    }
    return rho0;
}

// What ba_linearize needs per observation, without forming J_c or J_p: the corrected residual r, A (4x3: the
// loss-corrected derivative of the image coordinates with respect to the camera-frame point) and D = d(R x)/dw (3x3).
// Same arithmetic as residual_jacobian up to that point.  Returns rho(|r|^2).
// r (loss-corrected), Am = the loss-corrected 4x3 derivative with respect to the camera-frame point, vv = the vector the
// rotation derivative turns on (R x, or x on the first-order branch): J_c = [ A D | A ] with D = -[vv]x M.  Returns rho.
__device__ __forceinline__ double residual_av(const PosePre& pr, const double* x, const float4 uv, const Proj& P, double delta,
                                              double* __restrict__ r, double* __restrict__ Am, double* __restrict__ vv)
{
    double yr[3], y[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        yr[i] = pr.R[i * 3] * x[0] + pr.R[i * 3 + 1] * x[1] + pr.R[i * 3 + 2] * x[2];
        y[i] = yr[i] + pr.t[i];
    }
    double A[12], ul, vl, ur, vr;
    project_rows(P.l, y, ul, vl, A);
    project_rows(P.r, y, ur, vr, A + 6);
    r[0] = ul - (double)uv.x; r[1] = vl - (double)uv.y; r[2] = ur - (double)uv.z; r[3] = vr - (double)uv.w;
    double rho0, rho1;
    huber(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], delta, rho0, rho1);
    const double sw = sqrt(rho1);
#pragma unroll
    for (int i = 0; i < 3; i++) vv[i] = pr.small ? x[i] : yr[i];
#pragma unroll
    for (int i = 0; i < 12; i++) Am[i] = A[i] * sw;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] *= sw;
    return rho0;
}

// the same with D = -[vv]x M written out (3x3 row-major)
__device__ __forceinline__ double residual_ad(const PosePre& pr, const double* x, const float4 uv, const Proj& P, double delta,
                                              double* __restrict__ r, double* __restrict__ Am, double* __restrict__ D)
{
    double vv[3];
    const double rho0 = residual_av(pr, x, uv, P, delta, r, Am, vv);
    const double* M = pr.M;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        D[0 * 3 + j] = -(vv[1] * M[2 * 3 + j] - vv[2] * M[1 * 3 + j]);
        D[1 * 3 + j] = -(vv[2] * M[0 * 3 + j] - vv[0] * M[2 * 3 + j]);
        D[2 * 3 + j] = -(vv[0] * M[1 * 3 + j] - vv[1] * M[0 * 3 + j]);
    }
    return rho0;
}

// ---- compact rows --------------------------------------------------------------------------------------
// With A (4x3) the loss-corrected derivative of the four image coordinates with respect to the camera-frame point and
// r the corrected residual,
//     J_p = A R                              (4x3)
//     J_c = [ A D | A ],  D = -[v]x M        (4x6; v = R x, or x on the first-order branch; M = R Jr per camera)
// and everything downstream of ba_linearize needs A only through  G = A^T A  (3x3 symmetric) and  h = A^T r:
//     J_p^T J_p = R^T G R        J_p^T r = R^T h        J_p^T (J_c dc) = R^T G (w x v + dc_t),  w = M dc_rot
//     J_c^T J_p = T W~,   W~ = [ [v]x (G R) ; G R ],   T = blockdiag(M^T, I)      (D^T = M^T [v]x)
// ba_linearize therefore stores per observation the row [Gxx Gxy Gxz Gyy Gyz Gzz | hx hy hz | 0]: 80 B instead of the
// 320 B of r, J_c and J_p.  T is applied once per camera block after the per-point sums (ba_schur_reduce), w comes
// from ba_cam_update.
__device__ __forceinline__ void compact_v(const double* __restrict__ R, bool small, const double* __restrict__ x, double* __restrict__ v)
{
#pragma unroll
    for (int i = 0; i < 3; i++) v[i] = small ? x[i] : R[i * 3] * x[0] + R[i * 3 + 1] * x[1] + R[i * 3 + 2] * x[2];
}

// Z = G R (3x3 row-major), G symmetric as xx xy xz yy yz zz
__device__ __forceinline__ void compact_gr(const double* __restrict__ G, const double* __restrict__ R, double* __restrict__ z)
{
#pragma unroll
    for (int c = 0; c < 3; c++) {
        z[0 + c] = G[0] * R[c] + G[1] * R[3 + c] + G[2] * R[6 + c];
        z[3 + c] = G[1] * R[c] + G[3] * R[3 + c] + G[4] * R[6 + c];
        z[6 + c] = G[2] * R[c] + G[4] * R[3 + c] + G[5] * R[6 + c];
    }
}

// W~ (6x3 row-major): rows 3..5 = G R, rows 0..2 = [v]x (G R) = v x column
__device__ __forceinline__ void compact_wt(const double* __restrict__ G, const double* __restrict__ R, const double* __restrict__ v,
                                           double* __restrict__ w)
{
    compact_gr(G, R, w + 9);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double z0 = w[9 + c], z1 = w[12 + c], z2 = w[15 + c];
        w[0 + c] = v[1] * z2 - v[2] * z1;
        w[3 + c] = v[2] * z0 - v[0] * z2;
        w[6 + c] = v[0] * z1 - v[1] * z0;
    }
}

// inverse of a symmetric positive definite 3x3 (xx xy xz yy yz zz); ok=false when not positive definite
__device__ __forceinline__ bool sym3_inverse(const double* __restrict__ m, double* __restrict__ inv)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double m2 = a * d - b * b;
    const bool ok = (det > 0.0) && (a > 0.0) && (m2 > 0.0);
    const double id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
    inv[3] = (a * f - c * c) * id; inv[4] = (b * c - a * e) * id; inv[5] = m2 * id;
    return ok;
}

// the same with the reciprocal of the determinant from v_rcp_f64 and two Newton steps instead of the IEEE division sequence
// (div_scale / div_fmas / div_fixup: a dozen dependent instructions on the critical path of every pivot step of cr_invert).
// The result is within a few ulp of the correctly rounded one; the factor it feeds is an exact preconditioner either way.
__device__ __forceinline__ bool sym3_inverse_fast(const double* __restrict__ m, double* __restrict__ inv)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double m2 = a * d - b * b;
    const bool ok = (det > 0.0) && (a > 0.0) && (m2 > 0.0);
    double id = __builtin_amdgcn_rcp(det);
    id = id * (2.0 - det * id);
    id = id * (2.0 - det * id);
    inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
    inv[3] = (a * f - c * c) * id; inv[4] = (b * c - a * e) * id; inv[5] = m2 * id;
    return ok;
}

// Cholesky factor C = L L^T of a symmetric positive definite 3x3 (xx xy xz yy yz zz).  Out: mt = M = L^-T (upper
// triangular: m00 m01 m02 m11 m12 m22) and inv = C^-1 = M M^T.  ok = false when a pivot is not positive.
__device__ __forceinline__ bool sym3_chol_inverse(const double* __restrict__ m, double* __restrict__ mt, double* __restrict__ inv)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const double i00 = rsqrt(a);                 // 1 / l00
    const double l10 = b * i00, l20 = c * i00;
    const double p1 = d - l10 * l10;
    const double i11 = rsqrt(p1);                // 1 / l11
    const double l21 = (e - l20 * l10) * i11;
    const double p2 = f - l20 * l20 - l21 * l21;
    const double i22 = rsqrt(p2);                // 1 / l22
    const bool ok = (a > 0.0) && (p1 > 0.0) && (p2 > 0.0);
    // L^-1 = [[i00, 0, 0], [i10, i11, 0], [i20, i21, i22]]
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    mt[0] = i00; mt[1] = i10; mt[2] = i20; mt[3] = i11; mt[4] = i21; mt[5] = i22;
    inv[0] = i00 * i00 + i10 * i10 + i20 * i20;
    inv[1] = i10 * i11 + i20 * i21;
    inv[2] = i20 * i22;
    inv[3] = i11 * i11 + i21 * i21;
    inv[4] = i21 * i22;
    inv[5] = i22 * i22;
    return ok;
}

struct Publish {   // optional tail of a one-workgroup kernel: n <= 64 doubles to pinned host memory, then a sequence number;
                   // the slots [clear_first, clear_first + clear_n) of src (status words) are zeroed for the next iteration
    double* src;
    double* host_dst;
    unsigned long long* host_seq;
    unsigned long long seq;
    int n, clear_first, clear_n;
};

__device__ __forceinline__ void publish_tail(const Publish& pb)
{
    if (!pb.host_dst) return;   // uniform
    __syncthreads();            // the kernel's own result is among the published values
    const int t = threadIdx.x;
    if (t < pb.n) {
        pb.host_dst[t] = pb.src[t];
        if (t >= pb.clear_first && t < pb.clear_first + pb.clear_n) pb.src[t] = 0.0;
    }
    // only the first wave has written anything (n <= 64): the other waves of a 1024-lane kernel skip the system-scope
    // fence, whose cache write-back every wave would otherwise repeat
    if (t < 64) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __atomic_store_n(pb.host_seq, pb.seq, __ATOMIC_RELEASE);
}

}  // namespace soslam
