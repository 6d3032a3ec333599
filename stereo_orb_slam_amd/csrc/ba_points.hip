// ba_points.hip - structure-only bundle adjustment (every camera constant): BundleAdjuster::Optimize(n-1, n), the call the
// reference makes for every frame (/root/reference/src/slam.cpp:123).  The problem decouples into independent 3-variable
// blocks under ONE trust region, small enough (<= kPointsOnlyMax points) for a single workgroup:
//   ba_points_step    one LM iteration in one launch (linearise, damp, solve, project, evaluate the candidate, sum, publish)
//   ba_points_ls      one trial of the bounded problem's line search in one launch
//   ba_points_solve   the WHOLE solve in one launch: the trust-region controller (radius update, invalid-step rule, the
//                     termination tests) and Ceres' Armijo / cubic-interpolation line search run on the device, the host reads
//                     one record and the iteration log when the kernel has finished.
// The first two serve the host-side controller (run_lm: verbose / per-stage profiling / debug steps), the third is the product
// path of the per-frame call.
#include "ba_kernels.h"
#include "reduce.h"

namespace soslam {

namespace {

constexpr int kWave = 64;

// One LM iteration's arithmetic.  Leaves in fin[0..9]: cost at x, candidate cost, model cost change, |x+ - x|^2, |x|^2,
// gradient . step, max |gradient|, elimination failure flag, direction . gradient at the candidate, |step|_inf - readable by
// every lane on return.  red: 160 doubles of LDS.
template <int T>
__device__ __forceinline__ void points_step_pass(const PointsStepArgs& a, const Proj& P, const double* __restrict__ pts,
                                                 double* __restrict__ pts_out, const LmDiag& lm, const int init_scale, double* red, double* fin)
{
    double cost_x = 0.0, cost_c = 0.0, mcc = 0.0, st2 = 0.0, x2 = 0.0, gd = 0.0, gm = 0.0, dir_c = 0.0, dmax = 0.0;
    bool bad = false;
    for (uint32_t p = threadIdx.x; p < a.n_pt; p += T) {
        const double x[3] = {pts[3 * (size_t)p], pts[3 * (size_t)p + 1], pts[3 * (size_t)p + 2]};
        // linearisation at x: J_p = A R per observation
        double c[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        const uint32_t q0 = a.pt_start[p], q1 = a.pt_start[p + 1];
        for (uint32_t q = q0; q < q1; q++) {
            PosePre pr;
            pose_load(a.campre + kPoseStride * (size_t)a.q_cam[q], pr);
            double r[4], am[12], D[9];
            cost_x += residual_ad(pr, x, a.uv[a.pt_obs[q]], P, a.huber_delta, r, am, D);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                double j[3];
#pragma unroll
                for (int k = 0; k < 3; k++) j[k] = am[i * 3] * pr.R[k] + am[i * 3 + 1] * pr.R[3 + k] + am[i * 3 + 2] * pr.R[6 + k];
                c[0] += j[0] * j[0]; c[1] += j[0] * j[1]; c[2] += j[0] * j[2]; c[3] += j[1] * j[1]; c[4] += j[1] * j[2]; c[5] += j[2] * j[2];
                g[0] += j[0] * r[i]; g[1] += j[1] * r[i]; g[2] += j[2] * r[i];
            }
        }
        double sc[3];
        if (init_scale) {
            sc[0] = a.jacobi ? 1.0 / (1.0 + sqrt(c[0])) : 1.0; sc[1] = a.jacobi ? 1.0 / (1.0 + sqrt(c[3])) : 1.0;
            sc[2] = a.jacobi ? 1.0 / (1.0 + sqrt(c[5])) : 1.0;
#pragma unroll
            for (int k = 0; k < 3; k++) a.sp[3 * (size_t)p + k] = sc[k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) sc[k] = a.sp[3 * (size_t)p + k];
        }
        const double lam[3] = {point_lambda(c[0], sc[0], lm), point_lambda(c[3], sc[1], lm), point_lambda(c[5], sc[2], lm)};
        const double m[6] = {c[0] + lam[0], c[1], c[2], c[3] + lam[1], c[4], c[5] + lam[2]};
        double ci[6];
        if (!sym3_inverse(m, ci)) bad = true;
        const double e[3] = {-(ci[0] * g[0] + ci[1] * g[1] + ci[2] * g[2]), -(ci[1] * g[0] + ci[3] * g[1] + ci[4] * g[2]),
                             -(ci[2] * g[0] + ci[4] * g[1] + ci[5] * g[2])};
        double xc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            xc[k] = fmin(fmax(x[k] + e[k], a.bound_lo), a.bound_hi);
            const double st = xc[k] - x[k];
            pts_out[3 * (size_t)p + k] = xc[k];
            a.dp[3 * (size_t)p + k] = e[k];
            a.gp[3 * (size_t)p + k] = g[k];
            mcc += 0.5 * (lam[k] * e[k] * e[k] - g[k] * e[k]);
            st2 += st * st;
            x2 += x[k] * x[k];
            gd += g[k] * e[k];
            gm = fmax(gm, fabs(g[k]));
            dmax = fmax(dmax, fabs(e[k]));
        }
#pragma unroll
        for (int k = 0; k < 6; k++) { a.C[6 * (size_t)p + k] = c[k]; a.Cinv[6 * (size_t)p + k] = ci[k]; }
        // the candidate's cost, and direction . gradient there: the first trial of the bounded problem's line search (step size
        // 1) is this candidate, so the search starts without a launch of its own
        for (uint32_t q = q0; q < q1; q++) {
            PosePre pr;
            pose_load(a.campre + kPoseStride * (size_t)a.q_cam[q], pr);
            double r[4], am[12], D[9];
            cost_c += residual_ad(pr, xc, a.uv[a.pt_obs[q]], P, a.huber_delta, r, am, D);
            double u[3];   // J_p dp = A R dp
#pragma unroll
            for (int i = 0; i < 3; i++) u[i] = pr.R[i * 3] * e[0] + pr.R[i * 3 + 1] * e[1] + pr.R[i * 3 + 2] * e[2];
#pragma unroll
            for (int i = 0; i < 4; i++) dir_c += r[i] * (am[i * 3] * u[0] + am[i * 3 + 1] * u[1] + am[i * 3 + 2] * u[2]);
        }
    }
    cost_x = wave_sum(cost_x); cost_c = wave_sum(cost_c); mcc = wave_sum(mcc); st2 = wave_sum(st2); x2 = wave_sum(x2);
    gd = wave_sum(gd); gm = wave_max(gm); dir_c = wave_sum(dir_c); dmax = wave_max(dmax);
    const double badf = wave_max(bad ? 1.0 : 0.0);
    if (threadIdx.x % kWave == 0) {
        double* o = red + 10 * (threadIdx.x / kWave);
        o[0] = cost_x; o[1] = cost_c; o[2] = mcc; o[3] = st2; o[4] = x2; o[5] = gd; o[6] = gm; o[7] = badf; o[8] = dir_c; o[9] = dmax;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const int k = threadIdx.x;
        double v = 0.0;
        for (int w = 0; w < T / kWave; w++) v = (k == 6 || k == 7 || k == 9) ? fmax(v, red[10 * w + k]) : v + red[10 * w + k];
        if (k < 2) v *= 0.5;
        fin[k] = v;
    }
    __syncthreads();
}

// One line-search trial: candidate x+ = clamp(x + step dp) into pts_out; fin[0..3] = its cost, direction . gradient there,
// |x+ - x|^2, |dp|_inf - readable by every lane on return.  red: 64 doubles of LDS.
template <int T>
__device__ __forceinline__ void points_ls_pass(const PointsStepArgs& a, const Proj& P, const double* __restrict__ pts,
                                               double* __restrict__ pts_out, const double step, double* red, double* fin)
{
    double rho = 0.0, dir = 0.0, st2 = 0.0, dmax = 0.0;
    for (uint32_t p = threadIdx.x; p < a.n_pt; p += T) {
        double x[3], e[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double x0 = pts[3 * (size_t)p + k];
            e[k] = a.dp[3 * (size_t)p + k];
            x[k] = fmin(fmax(x0 + step * e[k], a.bound_lo), a.bound_hi);   // ParameterBlock::Plus projects onto the bounds
            pts_out[3 * (size_t)p + k] = x[k];
            st2 += (x[k] - x0) * (x[k] - x0);
            dmax = fmax(dmax, fabs(e[k]));
        }
        for (uint32_t q = a.pt_start[p]; q < a.pt_start[p + 1]; q++) {
            PosePre pr;
            pose_load(a.campre + kPoseStride * (size_t)a.q_cam[q], pr);
            double r[4], am[12], D[9];
            rho += residual_ad(pr, x, a.uv[a.pt_obs[q]], P, a.huber_delta, r, am, D);
            double u[3];   // J_p dp = A R dp
#pragma unroll
            for (int i = 0; i < 3; i++) u[i] = pr.R[i * 3] * e[0] + pr.R[i * 3 + 1] * e[1] + pr.R[i * 3 + 2] * e[2];
#pragma unroll
            for (int i = 0; i < 4; i++) dir += r[i] * (am[i * 3] * u[0] + am[i * 3 + 1] * u[1] + am[i * 3 + 2] * u[2]);
        }
    }
    rho = wave_sum(rho); dir = wave_sum(dir); st2 = wave_sum(st2); dmax = wave_max(dmax);
    if (threadIdx.x % kWave == 0) { double* o = red + 4 * (threadIdx.x / kWave); o[0] = rho; o[1] = dir; o[2] = st2; o[3] = dmax; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0.0;
        for (int w = 0; w < T / kWave; w++) v = threadIdx.x == 3 ? fmax(v, red[4 * w + 3]) : v + red[4 * w + threadIdx.x];
        fin[threadIdx.x] = threadIdx.x == 0 ? 0.5 * v : v;
    }
    __syncthreads();
}

}  // namespace

// ---- structure-only LM iteration in one launch (see PointsStepArgs) -------------------------------------------------------
__global__ __launch_bounds__(1024) void ba_points_step_kernel(const PointsStepArgs a, const Proj P, const Publish pb)
{
    __shared__ double red[16 * 10];
    __shared__ double fin[10];
    points_step_pass<1024>(a, P, a.pts, a.pts_out, a.lm, a.init_scale, red, fin);
    if (threadIdx.x == 0) {
        double* sc = a.scal;
        sc[SC_COST_X] = fin[0]; *a.cost_x_out = fin[0]; sc[SC_CAND_COST] = fin[1]; sc[SC_MCC_PTS] = fin[2]; sc[SC_STEP2_PTS] = fin[3]; sc[SC_X2_PTS] = fin[4];
        sc[SC_GDOT_PTS] = fin[5]; sc[SC_GMAX_PTS] = fin[6]; sc[SC_STOP] = 0.0;
        sc[SC_MCC_CAM] = 0.0; sc[SC_STEP2_CAM] = 0.0; sc[SC_X2_CAM] = 0.0; sc[SC_GDOT_CAM] = 0.0; sc[SC_GMAX_CAM] = 0.0;
        sc[SC_LIN_ITERS] = 0.0; sc[SC_LIN_RESID] = 0.0; sc[SC_LIN_STATUS] = 0.0; sc[SC_SCHUR_STATUS] = fin[7];
        sc[SC_LS_COST] = fin[1]; sc[SC_LS_DIR] = fin[8]; sc[SC_LS_STEP2] = fin[3]; sc[SC_LS_DMAX] = fin[9];   // the search's trial at step size 1
        const double mc = fin[2], cand = fin[1];
        const bool ok = fin[7] == 0.0 && isfinite(mc) && isfinite(cand) && mc > 0.0 && cand <= a.x_cost + 1e-4 * fin[5];
        sc[SC_GATE] = (a.gate_enabled && ok && (a.x_cost - cand) / mc > a.min_relative_decrease) ? 1.0 : 0.0;
    }
    publish_tail(pb);
}


void launch_points_step(hipStream_t s, const PointsStepArgs& a, const Proj& P, double* pub_src, int n_pub, int clear_first, int clear_n,
                        double* host_dst, unsigned long long* host_seq, unsigned long long seq)
{
    hipLaunchKernelGGL(ba_points_step_kernel, dim3(1), dim3(1024), 0, s, a, P, Publish{pub_src, host_dst, host_seq, seq, n_pub, clear_first, clear_n});
}

// One line-search trial of a structure-only problem in ONE launch (what ba_ls_candidate, ba_pose_prepare, ba_ls_eval and
// ba_ls_sums do for the general case), published.
__global__ __launch_bounds__(1024) void ba_points_ls_kernel(const PointsStepArgs a, const Proj P, const double step, const Publish pb)
{
    __shared__ double red[16 * 4];
    __shared__ double fin[4];
    points_ls_pass<1024>(a, P, a.pts, a.pts_out, step, red, fin);
    static_assert(SC_LS_DIR == SC_LS_COST + 1 && SC_LS_STEP2 == SC_LS_COST + 2 && SC_LS_DMAX == SC_LS_COST + 3, "line-search slots");
    if (threadIdx.x < 4) a.scal[SC_LS_COST + threadIdx.x] = fin[threadIdx.x];
    publish_tail(pb);
}

void launch_points_ls(hipStream_t s, const PointsStepArgs& a, const Proj& P, double step, double* pub_src, int n_pub, double* host_dst,
                      unsigned long long* host_seq, unsigned long long seq)
{
    hipLaunchKernelGGL(ba_points_ls_kernel, dim3(1), dim3(1024), 0, s, a, P, step, Publish{pub_src, host_dst, host_seq, seq, n_pub, 0, 0});
}

// ---- the whole structure-only solve in one launch --------------------------------------------------------------------------
// The polynomial arithmetic of Ceres' line search (FindInterpolatingPolynomial / MinimizePolynomial, CUBIC interpolation), for a
// single wave
namespace {

struct LsSample { double x, value, gradient; bool value_ok, gradient_ok; };

// line-search work area in LDS: the interpolation system (6 x 7), its solution, the derivative, the coefficient count, the
// root iteration's iterates and the roots' real parts
constexpr int kLsLd = 7, kLsMat = 0, kLsCoef = 42, kLsDer = 48, kLsNc = 54, kLsZre = 60, kLsZim = 64, kLsRoot = 68, kLsWork = 72;

__device__ double ls_ipow(double x, int n) { double v = 1.0; for (int i = 0; i < n; i++) v *= x; return v; }

__device__ double ls_polyval(const double* c, int n, double x)   // c[0] x^(n-1) + ... + c[n-1]
{
    double v = 0.0;
    for (int i = 0; i < n; i++) v = v * x + c[i];
    return v;
}

// polynomial through the samples' values and gradients (highest power first); returns the coefficient count
__device__ int ls_interpolate(const LsSample* smp, int ns, double* w)
{
    double* a = w + kLsMat;
    double* c = w + kLsCoef;
    int nc = 0;
    for (int s = 0; s < ns; s++) nc += (smp[s].value_ok ? 1 : 0) + (smp[s].gradient_ok ? 1 : 0);
    const int deg = nc - 1;
    for (int i = 0; i < 6 * kLsLd; i++) a[i] = 0.0;
    int row = 0;
    for (int s = 0; s < ns; s++) {
        if (smp[s].value_ok) {
            for (int j = 0; j <= deg; j++) a[row * kLsLd + j] = ls_ipow(smp[s].x, deg - j);
            a[row * kLsLd + nc] = smp[s].value; row++;
        }
        if (smp[s].gradient_ok) {
            for (int j = 0; j < deg; j++) a[row * kLsLd + j] = (deg - j) * ls_ipow(smp[s].x, deg - j - 1);
            a[row * kLsLd + nc] = smp[s].gradient; row++;
        }
    }
    for (int k = 0; k < nc; k++) {   // elimination with row pivoting
        int piv = k;
        for (int r = k + 1; r < nc; r++) if (fabs(a[r * kLsLd + k]) > fabs(a[piv * kLsLd + k])) piv = r;
        if (piv != k) for (int j = 0; j <= nc; j++) { const double t = a[k * kLsLd + j]; a[k * kLsLd + j] = a[piv * kLsLd + j]; a[piv * kLsLd + j] = t; }
        if (a[k * kLsLd + k] == 0.0) continue;
        for (int r = k + 1; r < nc; r++) {
            const double f = a[r * kLsLd + k] / a[k * kLsLd + k];
            for (int j = k; j <= nc; j++) a[r * kLsLd + j] -= f * a[k * kLsLd + j];
        }
    }
    for (int k = nc - 1; k >= 0; k--) {
        double v = a[k * kLsLd + nc];
        for (int j = k + 1; j < nc; j++) v -= a[k * kLsLd + j] * c[j];
        c[k] = a[k * kLsLd + k] != 0.0 ? v / a[k * kLsLd + k] : 0.0;
    }
    return nc;
}

struct Cplx { double re, im; };
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return Cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// 1 / b and a / b for the root iteration: the hardware reciprocal and one Newton step (the iteration corrects itself; what
// limits the roots is the polynomial's evaluation, not these quotients)
__device__ __forceinline__ double fast_rcp(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    return r * (2.0 - d * r);
}
__device__ __forceinline__ Cplx cinv(Cplx b)
{
    const double s = fmax(fabs(b.re), fabs(b.im));   // scaled: |b|^2 neither overflows nor vanishes
    const double is = fast_rcp(s), br = b.re * is, bi = b.im * is;
    const double id = fast_rcp(br * br + bi * bi) * is;
    return Cplx{br * id, -bi * id};
}

// Real parts of all roots of c[0] x^(n-1) + ... (Ceres evaluates the polynomial at the real part of complex roots too), into
// w[kLsRoot..]; returns their number.  Called by the whole wave, uniformly: degrees 1 and 2 in closed form, degrees 3 and 4
// (the derivative of the three-sample interpolant) by the Aberth-Ehrlich iteration with ONE ROOT PER LANE, all roots advanced
// together from the previous sweep's values.
__device__ int ls_root_real_parts(const double* c, int n, double* w, const int lane)
{
    double* re = w + kLsRoot;
    while (n > 1 && c[0] == 0.0) { c++; n--; }
    const int deg = n - 1;
    if (deg <= 0) return 0;
    if (deg <= 2) {
        if (lane == 0) {
            if (deg == 1) {
                re[0] = -c[1] / c[0];
            } else {
                const double a = c[0], b = c[1], cc = c[2], D = b * b - 4 * a * cc, sD = sqrt(fabs(D));
                if (D >= 0) {
                    if (b >= 0) { re[0] = (-b - sD) / (2.0 * a); re[1] = (2.0 * cc) / (-b - sD); }
                    else { re[0] = (2.0 * cc) / (-b + sD); re[1] = (-b + sD) / (2.0 * a); }
                } else {
                    re[0] = re[1] = -b / (2.0 * a);
                }
            }
        }
        __syncthreads();
        return deg;
    }
    double* zr = w + kLsZre;
    double* zi = w + kLsZim;
    double m[5];   // monic coefficients, in registers (deg <= 4)
    const double ic0 = 1.0 / c[0];
#pragma unroll
    for (int i = 0; i < 5; i++) m[i] = i <= deg ? c[i] * ic0 : 0.0;
    double rad = 0.0;
#pragma unroll
    for (int i = 1; i < 5; i++) if (i <= deg) rad = fmax(rad, pow(fabs(m[i]), 1.0 / i));
    if (rad == 0.0) {
        if (lane < deg) re[lane] = 0.0;
        __syncthreads();
        return deg;
    }
    const bool mine = lane < deg;
    Cplx z{0.0, 0.0};
    {
        double sn, cs;
        sincos(2.0 * 3.14159265358979323846 * (mine ? lane : 0) / deg + 0.4, &sn, &cs);
        z = Cplx{rad * cs, rad * sn};
    }
    const double tol2 = (1e-15 * rad) * (1e-15 * rad);
    for (int it = 0; it < 200; it++) {
        if (mine) { zr[lane] = z.re; zi[lane] = z.im; }
        __syncthreads();
        double moved2 = 0.0;
        if (mine) {
            Cplx pv{1.0, 0.0}, dv{0.0, 0.0};
#pragma unroll
            for (int k = 1; k < 5; k++) {
                if (k <= deg) {
                    dv = cmul(dv, z); dv.re += pv.re; dv.im += pv.im;
                    pv = cmul(pv, z); pv.re += m[k];
                }
            }
            if (!(dv.re == 0.0 && dv.im == 0.0)) {
                const Cplx wq = cmul(pv, cinv(dv));
                Cplx sum{0.0, 0.0};
                for (int j = 0; j < deg; j++) {
                    const Cplx o{zr[j], zi[j]};
                    if (j == lane || (o.re == z.re && o.im == z.im)) continue;
                    const Cplx t = cinv(Cplx{z.re - o.re, z.im - o.im});
                    sum.re += t.re; sum.im += t.im;
                }
                const Cplx ws = cmul(wq, sum);
                const Cplx q{1.0 - ws.re, -ws.im};
                const Cplx corr = (q.re == 0.0 && q.im == 0.0) ? wq : cmul(wq, cinv(q));
                z.re -= corr.re; z.im -= corr.im;
                moved2 = corr.re * corr.re + corr.im * corr.im;
            }
        }
        __syncthreads();   // everyone has read the sweep's values
        if (!(wave_max(moved2) > tol2)) break;
    }
    if (mine) re[lane] = z.re;
    __syncthreads();
    return deg;
}

// Ceres MinimizePolynomial on [x_min, x_max]: midpoint, both ends, the derivative's roots inside.  Whole wave, uniformly; the
// coefficients are in w[kLsCoef..] (ls_interpolate, lane 0, and a barrier behind it)
__device__ double ls_minimize(int nc, double x_min, double x_max, double* w, const int lane)
{
    const double* c = w + kLsCoef;
    double best_x = 0.5 * (x_min + x_max), best = ls_polyval(c, nc, best_x);
    { const double v = ls_polyval(c, nc, x_min); if (v < best) { best = v; best_x = x_min; } }
    { const double v = ls_polyval(c, nc, x_max); if (v < best) { best = v; best_x = x_max; } }
    const int deg = nc - 1;
    if (deg <= 1) return best_x;
    double* d = w + kLsDer;
    if (lane < deg) d[lane] = (deg - lane) * c[lane];
    __syncthreads();
    const int nr = ls_root_real_parts(d, deg, w, lane);
    const double* re = w + kLsRoot;
    for (int i = 0; i < nr; i++) {
        const double r = re[i];
        if (r < x_min || r > x_max) continue;
        const double v = ls_polyval(c, nc, r);
        if (v < best) { best = v; best_x = r; }
    }
    __syncthreads();   // the work area is free again
    return best_x;
}

// ---- the resident kernel --------------------------------------------------------------------------------------------------------
// One CU's f64 pipes are what bounded the one-workgroup form (16 waves: ~15 us per iteration for 1 000 points), so the points are
// spread one per lane over single-wave workgroups - each on a SIMD of its own - and the workgroups meet once per pass at a grid
// barrier that is also the reduction: every workgroup leaves its partial sums as one record (write-through stores, drained, then
// an agent-scope add to a monotonic counter), polls the counter and sums all records in a fixed order.  All workgroups run the
// same controller on the same totals and so take the same decisions; workgroup 0 writes the log and the record.  Per-point
// state - x, the step, the linearisation (J^T J, J^T r) at x and at the last trial - stays in registers: an accepted trial's
// linearisation IS the next iteration's, so a pass evaluates the residual and its derivative once, not twice.
// Hand-off form: MI355X_MICROARCH.md, "Valid forms", table row 1 (sc1 payload stores, vmcnt(0), agent atomic add; sc1 poll
// by the wave that then loads the payload with sc1 loads).
constexpr int kRecDoubles = 16;   // one 128-B line per workgroup and parity
enum { RV_COST_X = 0, RV_COST_T, RV_MCC, RV_ST2, RV_X2, RV_GDOT, RV_DIR, RV_GMAX, RV_DMAX, RV_GMAX_T, RV_BAD, RV_VOTE, RV_COUNT };
static_assert(RV_COUNT <= kRecDoubles, "record fits its line");

struct PointLin { double cost, c[6], g[3]; };

// cost, J_p^T J_p and J_p^T r of one point at x (J_p = A R per observation)
__device__ __forceinline__ void point_evaluate(const PointsStepArgs& a, const Proj& P, uint32_t q0, uint32_t q1, const double* x, PointLin& o)
{
    o.cost = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) o.c[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) o.g[k] = 0.0;
    for (uint32_t q = q0; q < q1; q++) {
        PosePre pr;
        pose_load(a.campre + kPoseStride * (size_t)a.q_cam[q], pr);
        double r[4], am[12], vv[3];
        o.cost += residual_av(pr, x, a.uv[a.pt_obs[q]], P, a.huber_delta, r, am, vv);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double j[3];
#pragma unroll
            for (int k = 0; k < 3; k++) j[k] = am[i * 3] * pr.R[k] + am[i * 3 + 1] * pr.R[3 + k] + am[i * 3 + 2] * pr.R[6 + k];
            o.c[0] += j[0] * j[0]; o.c[1] += j[0] * j[1]; o.c[2] += j[0] * j[2]; o.c[3] += j[1] * j[1]; o.c[4] += j[1] * j[2]; o.c[5] += j[2] * j[2];
            o.g[0] += j[0] * r[i]; o.g[1] += j[1] * r[i]; o.g[2] += j[2] * r[i];
        }
    }
}

__device__ __forceinline__ double uniform_f64(double x)   // lane 0's value in every lane (through scalar registers)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ double lane_f64(double x, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}

// The grid barrier that is the reduction.  v: this lane's contributions (zeros on idle lanes); on return the totals over all
// points, identical bits in every lane of every workgroup.  false: the barrier timed out (a workgroup is missing) - leave.
__device__ __forceinline__ bool grid_sums(const PointsSolveCtl& c, unsigned long long& pass, double (&v)[RV_COUNT])
{
    const int lane = threadIdx.x;
    // this wave's sums / maxima, then value k from lane k
    double mine = 0.0;
#pragma unroll
    for (int k = 0; k < RV_COUNT; k++) {
        const double t = (k == RV_GMAX || k == RV_DMAX || k == RV_GMAX_T) ? wave_max(v[k]) : wave_sum(v[k]);
        if (lane == k) mine = t;
    }
    double* rec = c.sync_records + ((pass & 1) * (size_t)c.n_wg + blockIdx.x) * kRecDoubles;
    if (lane < RV_COUNT) __hip_atomic_store(rec + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pass++;
    const unsigned long long target = c.sync_base + pass * (unsigned long long)c.n_wg;
    bool ok = true;
    if (lane == 0) {
        __hip_atomic_fetch_add(c.sync_counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(c.sync_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && wall_clock64() - t0 > 200000000ll) { ok = false; break; }   // two seconds: give up
        }
    }
    ok = __builtin_amdgcn_readfirstlane(ok ? 1 : 0) != 0;
    if (!ok) return false;
    asm volatile("" ::: "memory");
    // lane = 16 g + k sums value k over the records g, g + 4, ...; then the four groups; fixed order everywhere
    const int k = lane & 15, g = lane >> 4;
    const bool is_max = k == RV_GMAX || k == RV_DMAX || k == RV_GMAX_T;
    double acc = 0.0;
    const double* recs = c.sync_records + ((pass - 1) & 1) * (size_t)c.n_wg * kRecDoubles;
    if (k < RV_COUNT) {
        for (int w = g; w < c.n_wg; w += 4) {
            const double t = __hip_atomic_load(recs + (size_t)w * kRecDoubles + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc = is_max ? fmax(acc, t) : acc + t;
        }
    }
    {
        const F64Pair p = swap_rows16(acc);       // groups (0,1) and (2,3)
        acc = is_max ? fmax(p.a, p.b) : p.a + p.b;
        const F64Pair q = swap_halves32(acc);     // the two pairs
        acc = is_max ? fmax(q.a, q.b) : q.a + q.b;
    }
#pragma unroll
    for (int i = 0; i < RV_COUNT; i++) v[i] = lane_f64(acc, i);
    return true;
}

struct LogEntry {   // soslam_ba_iteration, as the device writes it
    double cost, candidate_cost, model_cost_change, relative_decrease, radius, step_norm, gradient_max_norm;
    int32_t accepted, valid, linear_iterations, reserved;
};
static_assert(sizeof(LogEntry) == 8 * kPointsLogDoubles, "log entries are copied out as doubles");

}  // namespace

// What run_lm (ba_solver.hip) does from the host, restated for the device.  Ceres: TrustRegionMinimizer (radius, StepIsInvalid,
// the tolerance tests), on bounded problems DoLineSearch -> ArmijoLineSearch::DoSearch with CUBIC interpolation
// (FindInterpolatingPolynomial / MinimizePolynomial).  The controller's scalars are wave-uniform (they come from grid_sums);
// lane 0 alone runs the polynomial arithmetic of the line search (LDS work area) and, in workgroup 0, writes the log.
__global__ __launch_bounds__(64) void ba_points_solve_kernel(const PointsStepArgs a, const Proj P, const PointsSolveCtl c)
{
    __shared__ double lsw[kLsWork];
    constexpr double kSufficientDecrease = 1e-4, kMaxContraction = 1e-3, kMinContraction = 0.6, kMinStepSize = 1e-9;
    constexpr int kMaxLsIterations = 20;
    LogEntry* log = reinterpret_cast<LogEntry*>(c.log);
    const int lane = threadIdx.x;
    const bool writer = blockIdx.x == 0 && lane == 0;
    const uint32_t p = blockIdx.x * 64u + (uint32_t)lane;
    const bool live = p < a.n_pt;
    // this lane's point
    double x[3] = {0.0, 0.0, 0.0}, sp[3] = {1.0, 1.0, 1.0}, e[3] = {0.0, 0.0, 0.0}, xt[3] = {0.0, 0.0, 0.0};
    uint32_t q0 = 0, q1 = 0;
    if (live) {
        const double* px = c.pts[c.cur] + 3 * (size_t)p;
        x[0] = px[0]; x[1] = px[1]; x[2] = px[2];
        q0 = a.pt_start[p]; q1 = a.pt_start[p + 1];
        if (!c.init_scale) { sp[0] = a.sp[3 * (size_t)p]; sp[1] = a.sp[3 * (size_t)p + 1]; sp[2] = a.sp[3 * (size_t)p + 2]; }
    }
    PointLin lin{}, tl{};
    bool have_lin = false;
    // the controller
    int invalid_run = c.invalid_run;
    double radius = c.radius, decrease = c.decrease_factor, x_cost = c.x_cost;
    bool x_known = c.x_cost_known != 0, have_initial = false, last_accepted = false;
    int it = 0, n_log = 0, accepted = 0, ls_steps = 0, term = kPointsTermMaxIterations, error = 0;
    double initial_cost = 0.0;
    unsigned long long pass = 0;
    bool time_vote = false;
    const long long t_start = wall_clock64();
    while (true) {
        if (it >= c.max_it) { term = kPointsTermMaxIterations; break; }
        if (c.check && radius < c.min_radius) { term = kPointsTermMinRadius; break; }
        if (time_vote) { term = kPointsTermTime; break; }   // some workgroup's clock passed the limit during the last pass
        // ---- the step: linearise at x if that is not at hand, damp, solve, project, evaluate the candidate ----
        double v[RV_COUNT];
#pragma unroll
        for (int k = 0; k < RV_COUNT; k++) v[k] = 0.0;
        if (live) {
            if (!have_lin) {
                point_evaluate(a, P, q0, q1, x, lin);
                if (c.init_scale) {   // the Jacobi scaling belongs to the first linearisation of a problem
                    sp[0] = a.jacobi ? 1.0 / (1.0 + sqrt(lin.c[0])) : 1.0; sp[1] = a.jacobi ? 1.0 / (1.0 + sqrt(lin.c[3])) : 1.0;
                    sp[2] = a.jacobi ? 1.0 / (1.0 + sqrt(lin.c[5])) : 1.0;
                    a.sp[3 * (size_t)p] = sp[0]; a.sp[3 * (size_t)p + 1] = sp[1]; a.sp[3 * (size_t)p + 2] = sp[2];
                }
                v[RV_COST_X] = lin.cost;
            }
            const LmDiag lm{radius, c.lm_lo, c.lm_hi};
            const double lam[3] = {point_lambda(lin.c[0], sp[0], lm), point_lambda(lin.c[3], sp[1], lm), point_lambda(lin.c[5], sp[2], lm)};
            const double m[6] = {lin.c[0] + lam[0], lin.c[1], lin.c[2], lin.c[3] + lam[1], lin.c[4], lin.c[5] + lam[2]};
            double ci[6];
            if (!sym3_inverse(m, ci)) v[RV_BAD] = 1.0;
            e[0] = -(ci[0] * lin.g[0] + ci[1] * lin.g[1] + ci[2] * lin.g[2]);
            e[1] = -(ci[1] * lin.g[0] + ci[3] * lin.g[1] + ci[4] * lin.g[2]);
            e[2] = -(ci[2] * lin.g[0] + ci[4] * lin.g[1] + ci[5] * lin.g[2]);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                xt[k] = fmin(fmax(x[k] + e[k], a.bound_lo), a.bound_hi);
                const double st = xt[k] - x[k];
                v[RV_MCC] += 0.5 * (lam[k] * e[k] * e[k] - lin.g[k] * e[k]);
                v[RV_ST2] += st * st;
                v[RV_X2] += x[k] * x[k];
                v[RV_GDOT] += lin.g[k] * e[k];
                v[RV_GMAX] = fmax(v[RV_GMAX], fabs(lin.g[k]));
                v[RV_DMAX] = fmax(v[RV_DMAX], fabs(e[k]));
            }
            point_evaluate(a, P, q0, q1, xt, tl);
            v[RV_COST_T] = tl.cost;
            v[RV_DIR] = tl.g[0] * e[0] + tl.g[1] * e[1] + tl.g[2] * e[2];   // direction . gradient at the candidate
            v[RV_GMAX_T] = fmax(fabs(tl.g[0]), fmax(fabs(tl.g[1]), fabs(tl.g[2])));
        }
        have_lin = true;
        if (c.max_ticks > 0 && wall_clock64() - t_start > c.max_ticks) v[RV_VOTE] = 1.0;
        if (!grid_sums(c, pass, v)) { error = 2; break; }
        time_vote = c.check && v[RV_VOTE] != 0.0;
        double cand = 0.5 * v[RV_COST_T], step_norm = sqrt(v[RV_ST2]), gmax_t = v[RV_GMAX_T];   // (gmax_t: max |gradient| at the trial)
        const double mcc = v[RV_MCC], x_norm = sqrt(v[RV_X2]), gdot = v[RV_GDOT], gmax = v[RV_GMAX], dir1 = v[RV_DIR], dmax = v[RV_DMAX];
        if (!x_known) { x_cost = 0.5 * v[RV_COST_X]; x_known = true; }
        if (!have_initial) {
            have_initial = true;
            initial_cost = x_cost;
            if (!isfinite(x_cost)) { error = 1; break; }
            if (writer) log[n_log] = LogEntry{x_cost, 0.0, 0.0, 0.0, radius, 0.0, gmax, 1, 1, 0, 0};
            n_log++;
            last_accepted = true;
        } else if (last_accepted) {
            if (writer) log[n_log - 1].gradient_max_norm = gmax;   // gradient at the point the last accepted step reached
        }
        if (c.check && gmax <= c.gradient_tolerance) { term = kPointsTermGradient; break; }
        it++;
        LogEntry en{x_cost, 0.0, mcc, 0.0, radius, 0.0, gmax, 0, 0, 0, 0};
        const bool lin_ok = v[RV_BAD] == 0.0 && isfinite(mcc) && isfinite(cand);
        if (!lin_ok || !(mcc > 0.0)) {
            if (writer) log[n_log] = en;
            n_log++;
            last_accepted = false;
            if (c.check && ++invalid_run >= 5) { term = kPointsTermInvalid; break; }
            radius *= 0.5;   // LevenbergMarquardtStrategy::StepIsInvalid
            continue;
        }
        invalid_run = 0;
        en.valid = 1;
        // bounded problem: Armijo search along the projected step before the candidate is judged; its first trial (step size 1)
        // is the candidate just evaluated
        if (c.constrained && !(cand <= x_cost + kSufficientDecrease * gdot)) {
            const LsSample initial{0.0, x_cost, gdot, true, true};
            LsSample previous{0.0, 0.0, 0.0, false, false};
            LsSample current{1.0, cand, dir1, isfinite(cand), isfinite(cand) && isfinite(dir1)};
            double ls_st2 = v[RV_ST2], ls_gmax = gmax_t;
            int iters = 0;
            bool found = true;
            while (!current.value_ok || current.value > x_cost + kSufficientDecrease * gdot * current.x) {
                iters++;
                if (iters >= kMaxLsIterations) { found = false; break; }
                const double lo = kMaxContraction * current.x, hi = kMinContraction * current.x;
                double step;
                if (!current.value_ok) {
                    step = fmin(fmax(current.x * 0.5, lo), hi);
                } else {
                    if (lane == 0) {
                        const LsSample smp[3] = {initial, current, previous};
                        lsw[kLsNc] = ls_interpolate(smp, previous.value_ok ? 3 : 2, lsw);
                    }
                    __syncthreads();
                    step = uniform_f64(ls_minimize((int)lsw[kLsNc], lo, hi, lsw, lane));
                }
                if (step * dmax < kMinStepSize) { found = false; break; }
                previous = current;
                double w[RV_COUNT];
#pragma unroll
                for (int k = 0; k < RV_COUNT; k++) w[k] = 0.0;
                if (live) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        xt[k] = fmin(fmax(x[k] + step * e[k], a.bound_lo), a.bound_hi);   // ParameterBlock::Plus projects onto the bounds
                        w[RV_ST2] += (xt[k] - x[k]) * (xt[k] - x[k]);
                    }
                    point_evaluate(a, P, q0, q1, xt, tl);
                    w[RV_COST_T] = tl.cost;
                    w[RV_DIR] = tl.g[0] * e[0] + tl.g[1] * e[1] + tl.g[2] * e[2];
                    w[RV_GMAX_T] = fmax(fabs(tl.g[0]), fmax(fabs(tl.g[1]), fabs(tl.g[2])));
                }
                if (!grid_sums(c, pass, w)) { error = 2; break; }
                current.x = step; current.value = 0.5 * w[RV_COST_T]; current.gradient = w[RV_DIR];
                current.value_ok = isfinite(current.value);
                current.gradient_ok = current.value_ok && isfinite(current.gradient);
                ls_st2 = w[RV_ST2];
                ls_gmax = w[RV_GMAX_T];
            }
            if (error) break;
            ls_steps += iters;
            if (!found) {
                // the search failed: Ceres leaves delta alone - the trial held in registers must be the full step again
                if (live) {
#pragma unroll
                    for (int k = 0; k < 3; k++) xt[k] = fmin(fmax(x[k] + e[k], a.bound_lo), a.bound_hi);
                    point_evaluate(a, P, q0, q1, xt, tl);
                }
            } else {
                cand = current.value;
                step_norm = sqrt(ls_st2);
                gmax_t = ls_gmax;
            }
        }
        en.candidate_cost = cand;
        en.step_norm = step_norm;
        if (c.check) {
            if (step_norm <= c.parameter_tolerance * (x_norm + c.parameter_tolerance)) {
                if (writer) log[n_log] = en;
                n_log++;
                term = kPointsTermParameter;
                break;
            }
            if (fabs(x_cost - cand) <= c.function_tolerance * x_cost) {
                if (writer) log[n_log] = en;
                n_log++;
                term = kPointsTermFunction;
                break;
            }
        }
        const double rel = (x_cost - cand) / mcc;
        en.relative_decrease = rel;
        if (rel > c.min_relative_decrease) {
            // the trial becomes x, its linearisation the next iteration's
#pragma unroll
            for (int k = 0; k < 3; k++) x[k] = xt[k];
            lin = tl;
            x_cost = cand;
            const double t = 2.0 * rel - 1.0;
            double f = 1.0 - t * t * t;
            if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            radius = fmin(c.max_radius, radius / f);
            decrease = 2.0;
            en.accepted = 1;
            en.cost = cand;
            en.gradient_max_norm = gmax_t;   // Ceres reports the gradient at the point an accepted step reaches
            accepted++;
        } else {
            radius /= decrease;
            decrease *= 2.0;
        }
        if (writer) log[n_log] = en;
        n_log++;
        last_accepted = en.accepted != 0;
    }
    // the points back (into the buffer they came from; whoever reads them is behind this kernel in stream order), then -
    // workgroup 0 - the record and the log into pinned host memory and the sequence number the host polls
    if (live) {
        double* px = c.pts[c.cur] + 3 * (size_t)p;
        px[0] = x[0]; px[1] = x[1]; px[2] = x[2];
    }
    if (blockIdx.x != 0) return;
    if (lane == 0) {
        double* r = c.host_record;
        r[PSV_RADIUS] = radius; r[PSV_DECREASE] = decrease; r[PSV_X_COST] = x_cost; r[PSV_CUR] = c.cur; r[PSV_INVALID_RUN] = invalid_run;
        r[PSV_ITERATIONS] = it; r[PSV_ACCEPTED] = accepted; r[PSV_TERMINATION] = term; r[PSV_LS_STEPS] = ls_steps;
        r[PSV_INITIAL_COST] = initial_cost; r[PSV_N_LOG] = n_log; r[PSV_ERROR] = error; r[PSV_HAVE_INITIAL] = have_initial ? 1.0 : 0.0;
        r[PSV_PASSES] = (double)pass;
    }
    __syncthreads();
    const double* lsrc = c.log;
    for (int i = lane; i < n_log * kPointsLogDoubles; i += 64) c.host_log[i] = lsrc[i];
    __threadfence_system();
    __syncthreads();
    if (lane == 0) __atomic_store_n(c.host_seq, c.seq, __ATOMIC_RELEASE);
}

void launch_points_solve(hipStream_t s, const PointsStepArgs& a, const Proj& P, const PointsSolveCtl& c)
{
    hipLaunchKernelGGL(ba_points_solve_kernel, dim3((unsigned)c.n_wg), dim3(64), 0, s, a, P, c);
}

}  // namespace soslam
