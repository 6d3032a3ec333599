/*
 * ba_oracle.c - CPU restatement of the reference's bundle-adjustment path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see ba_oracle.h).  PARITY UNPINNED: the
 * reference delegates this arithmetic to Ceres Solver, which is neither
 * vendored nor installed; semantics follow SURVEY.md Appendix A and are
 * cross-checked by oracle/gen_golden.py.
 *
 * What is restated, and from where:
 *   residual            /root/reference/src/reprojection_error.h:12-41
 *   derivative blocks   /root/reference/src/reprojection_error.h:53-62 (AutoDiff 4,6,3 => exact)
 *   loss                /root/reference/src/bundle_adjuster.cpp:100 (HuberLoss(1.0)) + Ceres corrector
 *   gauge               /root/reference/src/bundle_adjuster.cpp:113 (first pose constant)
 *   bounds              /root/reference/src/bundle_adjuster.cpp:104-108, /root/reference/src/params.h:44-47
 *   solver options      /root/reference/src/bundle_adjuster.cpp:14-36, /root/reference/src/params.h:34-41
 *   trust region        Ceres TRUST_REGION + LEVENBERG_MARQUARDT, SPARSE_SCHUR (direct) - Appendix A.3/A.4
 *
 * The linear algebra is explicit Schur elimination of the 3x3 point blocks
 * followed by an envelope (skyline) Cholesky of the reduced camera matrix - a
 * direct solve like the reference's CHOLMOD factorisation, exploiting the same
 * band structure so that CPU timings are a fair baseline.
 */
#define _POSIX_C_SOURCE 200809L
#include "ba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void oracle_ba_options_default(oracle_ba_options* o)
{
    o->max_iterations = 50;
    o->check_termination = 1;
    o->huber_delta = 1.0;
    o->lower_bound = -10000.0;
    o->upper_bound = 10000.0;
    o->initial_radius = 1e4;
    o->max_radius = 1e16;
    o->min_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->parameter_tolerance = 1e-8;
    o->function_tolerance = 1e-16;
    o->gradient_tolerance = 1e-16;
    o->jacobi_scaling = 1;
    o->num_threads = 1;
}

/* ---- per-observation arithmetic ----------------------------------------- */

/* y = R(w) x exactly as ceres::AngleAxisRotatePoint branches (Appendix A.1);
 * optionally dy/dw (3x3 row-major) and dy/dx (3x3) of the branch taken. */
static void rotate_point(const double* w, const double* x, double* y, double* dydw, double* dydx)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2 > DBL_EPSILON) {
        const double th = sqrt(th2), c = cos(th), s = sin(th), ith = 1.0 / th;
        const double k[3] = {w[0] * ith, w[1] * ith, w[2] * ith};
        const double kx[3] = {k[1] * x[2] - k[2] * x[1], k[2] * x[0] - k[0] * x[2], k[0] * x[1] - k[1] * x[0]};
        const double kd = k[0] * x[0] + k[1] * x[1] + k[2] * x[2];
        const double tmp = kd * (1.0 - c);
        for (int i = 0; i < 3; i++) y[i] = x[i] * c + kx[i] * s + k[i] * tmp;
        if (dydx) {
            /* c I + s [k]x + (1-c) k k^T */
            const double oc = 1.0 - c;
            dydx[0] = c + oc * k[0] * k[0];       dydx[1] = -s * k[2] + oc * k[0] * k[1]; dydx[2] = s * k[1] + oc * k[0] * k[2];
            dydx[3] = s * k[2] + oc * k[1] * k[0];  dydx[4] = c + oc * k[1] * k[1];       dydx[5] = -s * k[0] + oc * k[1] * k[2];
            dydx[6] = -s * k[1] + oc * k[2] * k[0]; dydx[7] = s * k[0] + oc * k[2] * k[1];  dydx[8] = c + oc * k[2] * k[2];
        }
        if (dydw) {
            for (int j = 0; j < 3; j++) {
                /* forward-mode derivative along e_j of the expression above */
                const double dth = k[j];
                double dk[3] = {-k[0] * k[j] * ith, -k[1] * k[j] * ith, -k[2] * k[j] * ith};
                dk[j] += ith;
                const double dkx[3] = {dk[1] * x[2] - dk[2] * x[1], dk[2] * x[0] - dk[0] * x[2], dk[0] * x[1] - dk[1] * x[0]};
                const double dkd = dk[0] * x[0] + dk[1] * x[1] + dk[2] * x[2];
                const double dtmp = dkd * (1.0 - c) + kd * s * dth;
                for (int i = 0; i < 3; i++)
                    dydw[i * 3 + j] = -x[i] * s * dth + dkx[i] * s + kx[i] * c * dth + dk[i] * tmp + k[i] * dtmp;
            }
        }
    } else {
        /* first-order branch: y = x + w x x */
        y[0] = x[0] + w[1] * x[2] - w[2] * x[1];
        y[1] = x[1] + w[2] * x[0] - w[0] * x[2];
        y[2] = x[2] + w[0] * x[1] - w[1] * x[0];
        if (dydx) {
            dydx[0] = 1.0;   dydx[1] = -w[2]; dydx[2] = w[1];
            dydx[3] = w[2];  dydx[4] = 1.0;   dydx[5] = -w[0];
            dydx[6] = -w[1]; dydx[7] = w[0];  dydx[8] = 1.0;
        }
        if (dydw) {
            dydw[0] = 0.0;   dydw[1] = x[2];  dydw[2] = -x[1];
            dydw[3] = -x[2]; dydw[4] = 0.0;   dydw[5] = x[0];
            dydw[6] = x[1];  dydw[7] = -x[0]; dydw[8] = 0.0;
        }
    }
}

static void project_rows(const double* P, const double* p, double* uv, double* a /* 2x3 or NULL */)
{
    const double d = P[8] * p[0] + P[9] * p[1] + P[10] * p[2] + P[11] * 1.0;
    const double inv = 1.0 / d;
    const double u = (P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[3] * 1.0) * inv;
    const double v = (P[4] * p[0] + P[5] * p[1] + P[6] * p[2] + P[7] * 1.0) * inv;
    uv[0] = u; uv[1] = v;
    if (a) {
        for (int i = 0; i < 3; i++) {
            a[i] = (P[i] - u * P[8 + i]) * inv;
            a[3 + i] = (P[4 + i] - v * P[8 + i]) * inv;
        }
    }
}

void oracle_ba_residual(const double* cam, const double* pt, const double* uv,
                        const double* pl, const double* pr, double* r)
{
    double y[3], l[2], q[2];
    rotate_point(cam, pt, y, NULL, NULL);
    y[0] += cam[3]; y[1] += cam[4]; y[2] += cam[5];
    project_rows(pl, y, l, NULL);
    project_rows(pr, y, q, NULL);
    r[0] = l[0] - uv[0]; r[1] = l[1] - uv[1]; r[2] = q[0] - uv[2]; r[3] = q[1] - uv[3];
}

void oracle_ba_residual_jacobian(const double* cam, const double* pt, const double* uv,
                                 const double* pl, const double* pr,
                                 double* r, double* jc, double* jp)
{
    double y[3], dydw[9], dydx[9], l[2], q[2], A[12];
    rotate_point(cam, pt, y, dydw, dydx);
    y[0] += cam[3]; y[1] += cam[4]; y[2] += cam[5];
    project_rows(pl, y, l, A);
    project_rows(pr, y, q, A + 6);
    r[0] = l[0] - uv[0]; r[1] = l[1] - uv[1]; r[2] = q[0] - uv[2]; r[3] = q[1] - uv[3];
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 3; j++) {
            jc[i * 6 + j] = A[i * 3 + 0] * dydw[0 * 3 + j] + A[i * 3 + 1] * dydw[1 * 3 + j] + A[i * 3 + 2] * dydw[2 * 3 + j];
            jc[i * 6 + 3 + j] = A[i * 3 + j];
            jp[i * 3 + j] = A[i * 3 + 0] * dydx[0 * 3 + j] + A[i * 3 + 1] * dydx[1 * 3 + j] + A[i * 3 + 2] * dydx[2 * 3 + j];
        }
    }
}

void oracle_huber(double s, double delta, double* rho)
{
    const double b = delta * delta;
    if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * delta * r - b;
        rho[1] = delta / r;
        if (rho[1] < DBL_MIN) rho[1] = DBL_MIN;
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

static double obs_cost(const double* cam, const double* pt, const float* uvf,
                       const double* pl, const double* pr, double delta)
{
    double uv[4] = {uvf[0], uvf[1], uvf[2], uvf[3]}, r[4], rho[3];
    oracle_ba_residual(cam, pt, uv, pl, pr, r);
    oracle_huber(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], delta, rho);
    return rho[0];
}

double oracle_ba_cost(uint32_t n_obs, const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                      const double* cams, const double* pts, const double* pl, const double* pr, double delta)
{
    double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
    for (int64_t k = 0; k < (int64_t)n_obs; k++)
        sum += obs_cost(cams + 6 * (size_t)obs_cam[k], pts + 3 * (size_t)obs_pt[k], obs_uv + 4 * (size_t)k, pl, pr, delta);
    return 0.5 * sum;
}

double oracle_ba_linearize(uint32_t n_obs, const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                           const double* cams, const double* pts, const double* pl, const double* pr,
                           const uint8_t* cam_fixed, double delta, double* r_out, double* jc_out, double* jp_out)
{
    double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
    for (int64_t k = 0; k < (int64_t)n_obs; k++) {
        const uint32_t c = obs_cam[k];
        const float* uvf = obs_uv + 4 * (size_t)k;
        double uv[4] = {uvf[0], uvf[1], uvf[2], uvf[3]}, r[4], jc[24], jp[12], rho[3];
        oracle_ba_residual_jacobian(cams + 6 * (size_t)c, pts + 3 * (size_t)obs_pt[k], uv, pl, pr, r, jc, jp);
        oracle_huber(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], delta, rho);
        sum += rho[0];
        /* Ceres corrector with rho'' <= 0: scale residual and Jacobian by sqrt(rho') (Appendix A.2) */
        const double w = sqrt(rho[1]);
        const int fixed = cam_fixed && cam_fixed[c];
        if (r_out) for (int i = 0; i < 4; i++) r_out[4 * (size_t)k + i] = w * r[i];
        if (jc_out) for (int i = 0; i < 24; i++) jc_out[24 * (size_t)k + i] = fixed ? 0.0 : w * jc[i];
        if (jp_out) for (int i = 0; i < 12; i++) jp_out[12 * (size_t)k + i] = w * jp[i];
    }
    return 0.5 * sum;
}

/* ---- workspace ----------------------------------------------------------- */

typedef struct ws_t {
    uint32_t n_cam, n_pt, n_obs, n_free;
    const uint32_t* obs_cam; const uint32_t* obs_pt; const float* uv;
    const double* pl; const double* pr; const uint8_t* fixed;
    const oracle_ba_options* opt;
    int32_t* free_idx;                 /* camera -> free index or -1 */
    uint32_t* pt_start; uint32_t* pt_obs;   /* observations of each point, camera ascending */
    uint32_t* cam_start; uint32_t* cam_obs; /* observations of each camera */
    double *r, *jc, *jp, *W;           /* per observation: 4, 24, 12, 18 (W = Jc^T Jp, 6x3) */
    double *B, *gc;                    /* per free camera: 36, 6 */
    double *C, *gp, *Cinv;             /* per point: 6 (xx xy xz yy yz zz), 3, 6 */
    double *sc, *sp;                   /* Jacobi scales: n_free*6, n_pt*3 */
    double *lc, *lp;                   /* unscaled damping: n_free*6, n_pt*3 */
    double *S, *rhs; int32_t* first;   /* dense reduced system (6F)^2, rhs, envelope start per row */
    double *dc, *dp;                   /* step: n_cam*6, n_pt*3 */
    double *xc, *xp;                   /* candidate */
} ws_t;

static void* xcalloc(size_t n, size_t sz)
{
    void* p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "ba_oracle: out of memory (%zu x %zu)\n", n, sz); abort(); }
    return p;
}

static void ws_free(ws_t* w)
{
    free(w->free_idx); free(w->pt_start); free(w->pt_obs); free(w->cam_start); free(w->cam_obs);
    free(w->r); free(w->jc); free(w->jp); free(w->W); free(w->B); free(w->gc); free(w->C); free(w->gp);
    free(w->Cinv); free(w->sc); free(w->sp); free(w->lc); free(w->lp); free(w->S); free(w->rhs);
    free(w->first); free(w->dc); free(w->dp); free(w->xc); free(w->xp);
    memset(w, 0, sizeof *w);
}

static int ws_init(ws_t* w, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                   const uint32_t* obs_cam, const uint32_t* obs_pt, const float* uv,
                   const double* pl, const double* pr, const uint8_t* fixed, const oracle_ba_options* opt)
{
    memset(w, 0, sizeof *w);
    w->n_cam = n_cam; w->n_pt = n_pt; w->n_obs = n_obs;
    w->obs_cam = obs_cam; w->obs_pt = obs_pt; w->uv = uv; w->pl = pl; w->pr = pr; w->fixed = fixed; w->opt = opt;
    for (uint32_t k = 0; k < n_obs; k++) if (obs_cam[k] >= n_cam || obs_pt[k] >= n_pt) return -1;
    w->free_idx = (int32_t*)xcalloc(n_cam, sizeof(int32_t));
    uint32_t nf = 0;
    for (uint32_t c = 0; c < n_cam; c++) w->free_idx[c] = (fixed && fixed[c]) ? -1 : (int32_t)nf++;
    w->n_free = nf;
    /* CSR by point and by camera (counting sort keeps input order inside each list; input is
       frame-major, so per-point lists come out camera-ascending) */
    w->pt_start = (uint32_t*)xcalloc((size_t)n_pt + 1, sizeof(uint32_t));
    w->cam_start = (uint32_t*)xcalloc((size_t)n_cam + 1, sizeof(uint32_t));
    w->pt_obs = (uint32_t*)xcalloc(n_obs, sizeof(uint32_t));
    w->cam_obs = (uint32_t*)xcalloc(n_obs, sizeof(uint32_t));
    for (uint32_t k = 0; k < n_obs; k++) { w->pt_start[obs_pt[k] + 1]++; w->cam_start[obs_cam[k] + 1]++; }
    for (uint32_t p = 0; p < n_pt; p++) w->pt_start[p + 1] += w->pt_start[p];
    for (uint32_t c = 0; c < n_cam; c++) w->cam_start[c + 1] += w->cam_start[c];
    {
        uint32_t* fp = (uint32_t*)xcalloc(n_pt, sizeof(uint32_t));
        uint32_t* fc = (uint32_t*)xcalloc(n_cam, sizeof(uint32_t));
        for (uint32_t k = 0; k < n_obs; k++) {
            w->pt_obs[w->pt_start[obs_pt[k]] + fp[obs_pt[k]]++] = k;
            w->cam_obs[w->cam_start[obs_cam[k]] + fc[obs_cam[k]]++] = k;
        }
        free(fp); free(fc);
    }
    w->r = (double*)xcalloc((size_t)n_obs * 4, sizeof(double));
    w->jc = (double*)xcalloc((size_t)n_obs * 24, sizeof(double));
    w->jp = (double*)xcalloc((size_t)n_obs * 12, sizeof(double));
    w->W = (double*)xcalloc((size_t)n_obs * 18, sizeof(double));
    w->B = (double*)xcalloc((size_t)nf * 36, sizeof(double));
    w->gc = (double*)xcalloc((size_t)nf * 6, sizeof(double));
    w->C = (double*)xcalloc((size_t)n_pt * 6, sizeof(double));
    w->gp = (double*)xcalloc((size_t)n_pt * 3, sizeof(double));
    w->Cinv = (double*)xcalloc((size_t)n_pt * 6, sizeof(double));
    w->sc = (double*)xcalloc((size_t)nf * 6, sizeof(double));
    w->sp = (double*)xcalloc((size_t)n_pt * 3, sizeof(double));
    w->lc = (double*)xcalloc((size_t)nf * 6, sizeof(double));
    w->lp = (double*)xcalloc((size_t)n_pt * 3, sizeof(double));
    w->S = (double*)xcalloc((size_t)nf * 6 * nf * 6, sizeof(double));
    w->rhs = (double*)xcalloc((size_t)nf * 6, sizeof(double));
    w->first = (int32_t*)xcalloc((size_t)nf * 6, sizeof(int32_t));
    w->dc = (double*)xcalloc((size_t)n_cam * 6, sizeof(double));
    w->dp = (double*)xcalloc((size_t)n_pt * 3, sizeof(double));
    w->xc = (double*)xcalloc((size_t)n_cam * 6, sizeof(double));
    w->xp = (double*)xcalloc((size_t)n_pt * 3, sizeof(double));
    return 0;
}

/* J^T J blocks and J^T r from the corrected per-observation blocks */
static void build_normal(ws_t* w)
{
    const uint32_t n_pt = w->n_pt, n_cam = w->n_cam;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)w->n_obs; k++) {
        const double* jc = w->jc + 24 * (size_t)k; const double* jp = w->jp + 12 * (size_t)k;
        double* W = w->W + 18 * (size_t)k;
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 3; b++)
                W[a * 3 + b] = jc[0 * 6 + a] * jp[0 * 3 + b] + jc[1 * 6 + a] * jp[1 * 3 + b] +
                               jc[2 * 6 + a] * jp[2 * 3 + b] + jc[3 * 6 + a] * jp[3 * 3 + b];
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t p = 0; p < (int64_t)n_pt; p++) {
        double C[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
        for (uint32_t q = w->pt_start[p]; q < w->pt_start[p + 1]; q++) {
            const uint32_t k = w->pt_obs[q];
            const double* jp = w->jp + 12 * (size_t)k; const double* r = w->r + 4 * (size_t)k;
            for (int i = 0; i < 4; i++) {
                const double a = jp[i * 3], b = jp[i * 3 + 1], c = jp[i * 3 + 2];
                C[0] += a * a; C[1] += a * b; C[2] += a * c; C[3] += b * b; C[4] += b * c; C[5] += c * c;
                g[0] += a * r[i]; g[1] += b * r[i]; g[2] += c * r[i];
            }
        }
        memcpy(w->C + 6 * (size_t)p, C, sizeof C);
        memcpy(w->gp + 3 * (size_t)p, g, sizeof g);
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t c = 0; c < (int64_t)n_cam; c++) {
        const int32_t f = w->free_idx[c];
        if (f < 0) continue;
        double B[36], g[6];
        memset(B, 0, sizeof B); memset(g, 0, sizeof g);
        for (uint32_t q = w->cam_start[c]; q < w->cam_start[c + 1]; q++) {
            const uint32_t k = w->cam_obs[q];
            const double* jc = w->jc + 24 * (size_t)k; const double* r = w->r + 4 * (size_t)k;
            for (int i = 0; i < 4; i++)
                for (int a = 0; a < 6; a++) {
                    g[a] += jc[i * 6 + a] * r[i];
                    for (int b = a; b < 6; b++) B[a * 6 + b] += jc[i * 6 + a] * jc[i * 6 + b];
                }
        }
        for (int a = 0; a < 6; a++) for (int b = 0; b < a; b++) B[a * 6 + b] = B[b * 6 + a];
        memcpy(w->B + 36 * (size_t)f, B, sizeof B);
        memcpy(w->gc + 6 * (size_t)f, g, sizeof g);
    }
}

static const int CDIAG[3] = {0, 3, 5};

static void compute_scale(ws_t* w)
{
    const int on = w->opt->jacobi_scaling;
    for (uint32_t f = 0; f < w->n_free; f++)
        for (int a = 0; a < 6; a++)
            w->sc[6 * (size_t)f + a] = on ? 1.0 / (1.0 + sqrt(w->B[36 * (size_t)f + a * 7])) : 1.0;
    for (uint32_t p = 0; p < w->n_pt; p++)
        for (int a = 0; a < 3; a++)
            w->sp[3 * (size_t)p + a] = on ? 1.0 / (1.0 + sqrt(w->C[6 * (size_t)p + CDIAG[a]])) : 1.0;
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* unscaled damping equivalent to Ceres' D^2 = clamp(diag(Js^T Js)) / radius on the scaled Jacobian Js = J diag(s) */
static void compute_damping(ws_t* w, double radius)
{
    const double lo = w->opt->min_lm_diagonal, hi = w->opt->max_lm_diagonal;
    for (uint32_t f = 0; f < w->n_free; f++)
        for (int a = 0; a < 6; a++) {
            const double s = w->sc[6 * (size_t)f + a], s2 = s * s;
            w->lc[6 * (size_t)f + a] = clampd(s2 * w->B[36 * (size_t)f + a * 7], lo, hi) / (radius * s2);
        }
    for (uint32_t p = 0; p < w->n_pt; p++)
        for (int a = 0; a < 3; a++) {
            const double s = w->sp[3 * (size_t)p + a], s2 = s * s;
            w->lp[3 * (size_t)p + a] = clampd(s2 * w->C[6 * (size_t)p + CDIAG[a]], lo, hi) / (radius * s2);
        }
}

/* inverse of the symmetric 3x3 (xx xy xz yy yz zz); returns 0 if not positive definite */
static int sym3_inverse(const double* m, double* inv)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (!(det > 0.0) || !(a > 0.0) || !(a * d - b * b > 0.0)) return 0;
    const double id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
    inv[3] = (a * f - c * c) * id; inv[4] = (b * c - a * e) * id; inv[5] = (a * d - b * b) * id;
    return 1;
}

/*
 * Reduced camera system for the point range [p_begin, p_end): S (upper blocks by free index) gets
 *   -sum_p Y W^T, rhs gets +sum_p Y g_p, with Y = W Cinv.  Camera blocks B, damping and -g_c are
 *   added by the caller.  Row-parallel: the row of the smaller free index owns each block pair.
 */
static int schur_accumulate(ws_t* w, uint32_t p_begin, uint32_t p_end, double* S, double* rhs, int32_t* first_blk)
{
    const size_t n6 = (size_t)w->n_free * 6;
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(| : bad)
    for (int64_t p = p_begin; p < (int64_t)p_end; p++) {
        double m[6];
        memcpy(m, w->C + 6 * (size_t)p, sizeof m);
        m[0] += w->lp[3 * (size_t)p]; m[3] += w->lp[3 * (size_t)p + 1]; m[5] += w->lp[3 * (size_t)p + 2];
        if (!sym3_inverse(m, w->Cinv + 6 * (size_t)p)) bad |= 1;
    }
    if (bad) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c = 0; c < (int64_t)w->n_cam; c++) {
        const int32_t fi = w->free_idx[c];
        if (fi < 0) continue;
        int32_t minj = first_blk ? first_blk[fi] : fi;
        for (uint32_t qa = w->cam_start[c]; qa < w->cam_start[c + 1]; qa++) {
            const uint32_t ka = w->cam_obs[qa], p = w->obs_pt[ka];
            if (p < p_begin || p >= p_end) continue;
            const double* Wa = w->W + 18 * (size_t)ka; const double* ci = w->Cinv + 6 * (size_t)p;
            const double* g = w->gp + 3 * (size_t)p;
            double Y[18];
            for (int a = 0; a < 6; a++) {
                const double x = Wa[a * 3], y = Wa[a * 3 + 1], z = Wa[a * 3 + 2];
                Y[a * 3] = x * ci[0] + y * ci[1] + z * ci[2];
                Y[a * 3 + 1] = x * ci[1] + y * ci[3] + z * ci[4];
                Y[a * 3 + 2] = x * ci[2] + y * ci[4] + z * ci[5];
                rhs[6 * (size_t)fi + a] += Y[a * 3] * g[0] + Y[a * 3 + 1] * g[1] + Y[a * 3 + 2] * g[2];
            }
            for (uint32_t qb = w->pt_start[p]; qb < w->pt_start[p + 1]; qb++) {
                const uint32_t kb = w->pt_obs[qb];
                const int32_t fj = w->free_idx[w->obs_cam[kb]];
                if (fj < fi) continue; /* fixed (-1) or owned by the other row */
                const double* Wb = w->W + 18 * (size_t)kb;
                double* blk = S + (6 * (size_t)fi) * n6 + 6 * (size_t)fj;
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 6; b++)
                        blk[a * n6 + b] -= Y[a * 3] * Wb[b * 3] + Y[a * 3 + 1] * Wb[b * 3 + 1] + Y[a * 3 + 2] * Wb[b * 3 + 2];
            }
        }
        (void)minj;
    }
    return 0;
}

/* envelope of the reduced matrix: first co-visible free camera (column block) of each free camera */
static void compute_envelope(ws_t* w)
{
    const uint32_t nf = w->n_free;
    int32_t* fb = (int32_t*)xcalloc(nf, sizeof(int32_t));
    for (uint32_t f = 0; f < nf; f++) fb[f] = (int32_t)f;
    for (uint32_t p = 0; p < w->n_pt; p++) {
        int32_t lo = -1;
        for (uint32_t q = w->pt_start[p]; q < w->pt_start[p + 1]; q++) {
            int32_t f = w->free_idx[w->obs_cam[w->pt_obs[q]]];
            if (f >= 0 && (lo < 0 || f < lo)) lo = f;
        }
        if (lo < 0) continue;
        for (uint32_t q = w->pt_start[p]; q < w->pt_start[p + 1]; q++) {
            int32_t f = w->free_idx[w->obs_cam[w->pt_obs[q]]];
            if (f >= 0 && lo < fb[f]) fb[f] = lo;
        }
    }
    /* a skyline factor needs a monotone-free envelope only per row; rows of one block share it */
    for (uint32_t f = 0; f < nf; f++) for (int a = 0; a < 6; a++) w->first[6 * (size_t)f + a] = 6 * fb[f];
    free(fb);
}

/* in-place envelope Cholesky of the lower triangle, then solve L L^T x = b (x overwrites b) */
static int skyline_cholesky_solve(double* S, const int32_t* first, size_t n, double* b)
{
    for (size_t i = 0; i < n; i++) {
        double* Li = S + i * n;
        const size_t fi = (size_t)first[i];
        for (size_t j = fi; j <= i; j++) {
            const double* Lj = S + j * n;
            const size_t fj = (size_t)first[j];
            const size_t k0 = fi > fj ? fi : fj;
            double sum = Li[j];
            for (size_t k = k0; k < j; k++) sum -= Li[k] * Lj[k];
            if (j < i) Li[j] = sum / Lj[j];
            else {
                if (!(sum > 0.0)) return -1;
                Li[i] = sqrt(sum);
            }
        }
    }
    for (size_t i = 0; i < n; i++) {
        const double* Li = S + i * n;
        double sum = b[i];
        for (size_t k = (size_t)first[i]; k < i; k++) sum -= Li[k] * b[k];
        b[i] = sum / Li[i];
    }
    for (size_t ii = n; ii-- > 0;) {
        const double* Li = S + ii * n;
        b[ii] /= Li[ii];
        const double x = b[ii];
        for (size_t k = (size_t)first[ii]; k < ii; k++) b[k] -= Li[k] * x;
    }
    return 0;
}

/*
 * Solve (H + Lambda) Delta = -g by Schur elimination of the points.
 * keep_dense: leave the assembled (unfactored) S, both triangles, in s_copy and rhs in rhs_copy.
 */
static int solve_step(ws_t* w, double radius, double* s_copy, double* rhs_copy)
{
    const uint32_t nf = w->n_free;
    const size_t n6 = (size_t)nf * 6;
    compute_damping(w, radius);
    memset(w->S, 0, sizeof(double) * n6 * n6);
    memset(w->rhs, 0, sizeof(double) * n6);
    if (schur_accumulate(w, 0, w->n_pt, w->S, w->rhs, NULL) != 0) return -1;
    for (uint32_t f = 0; f < nf; f++) {
        for (int a = 0; a < 6; a++) {
            for (int b = a; b < 6; b++) w->S[(6 * (size_t)f + a) * n6 + 6 * (size_t)f + b] += w->B[36 * (size_t)f + a * 6 + b];
            w->S[(6 * (size_t)f + a) * n6 + 6 * (size_t)f + a] += w->lc[6 * (size_t)f + a];
            w->rhs[6 * (size_t)f + a] -= w->gc[6 * (size_t)f + a];
        }
    }
    /* mirror the upper triangle (where the accumulation lives) into the lower one */
    for (size_t i = 0; i < n6; i++) {
        const size_t j0 = (size_t)w->first[i];
        for (size_t j = j0; j < i; j++) w->S[i * n6 + j] = w->S[j * n6 + i];
    }
    if (s_copy) {
        memcpy(s_copy, w->S, sizeof(double) * n6 * n6);
        for (size_t i = 0; i < n6; i++) for (size_t j = i + 1; j < n6; j++) s_copy[i * n6 + j] = s_copy[j * n6 + i];
    }
    if (rhs_copy) memcpy(rhs_copy, w->rhs, sizeof(double) * n6);
    if (n6 && skyline_cholesky_solve(w->S, w->first, n6, w->rhs) != 0) return -2;
    memset(w->dc, 0, sizeof(double) * 6 * (size_t)w->n_cam);
    for (uint32_t c = 0; c < w->n_cam; c++) {
        const int32_t f = w->free_idx[c];
        if (f >= 0) memcpy(w->dc + 6 * (size_t)c, w->rhs + 6 * (size_t)f, 6 * sizeof(double));
    }
    /* back-substitution: dp = -Cinv (g_p + sum_obs W^T dc) */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t p = 0; p < (int64_t)w->n_pt; p++) {
        double t[3] = {w->gp[3 * (size_t)p], w->gp[3 * (size_t)p + 1], w->gp[3 * (size_t)p + 2]};
        for (uint32_t q = w->pt_start[p]; q < w->pt_start[p + 1]; q++) {
            const uint32_t k = w->pt_obs[q];
            const double* d = w->dc + 6 * (size_t)w->obs_cam[k];
            const double* W = w->W + 18 * (size_t)k;
            for (int a = 0; a < 6; a++) { t[0] += W[a * 3] * d[a]; t[1] += W[a * 3 + 1] * d[a]; t[2] += W[a * 3 + 2] * d[a]; }
        }
        const double* ci = w->Cinv + 6 * (size_t)p;
        w->dp[3 * (size_t)p] = -(ci[0] * t[0] + ci[1] * t[1] + ci[2] * t[2]);
        w->dp[3 * (size_t)p + 1] = -(ci[1] * t[0] + ci[3] * t[1] + ci[4] * t[2]);
        w->dp[3 * (size_t)p + 2] = -(ci[2] * t[0] + ci[4] * t[1] + ci[5] * t[2]);
    }
    return 0;
}

/* Ceres: model_cost_change = -(J d)^T (r + J d / 2) on the corrected blocks */
static double model_cost_change(const ws_t* w)
{
    double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
    for (int64_t k = 0; k < (int64_t)w->n_obs; k++) {
        const double* jc = w->jc + 24 * (size_t)k; const double* jp = w->jp + 12 * (size_t)k;
        const double* r = w->r + 4 * (size_t)k;
        const double* d = w->dc + 6 * (size_t)w->obs_cam[k]; const double* e = w->dp + 3 * (size_t)w->obs_pt[k];
        for (int i = 0; i < 4; i++) {
            double m = jp[i * 3] * e[0] + jp[i * 3 + 1] * e[1] + jp[i * 3 + 2] * e[2];
            for (int a = 0; a < 6; a++) m += jc[i * 6 + a] * d[a];
            sum -= m * (r[i] + 0.5 * m);
        }
    }
    return sum;
}

static void make_candidate(ws_t* w, const double* cams, const double* pts, double* step_norm, double* x_norm)
{
    double sn = 0.0, xn = 0.0;
    for (uint32_t c = 0; c < w->n_cam; c++)
        for (int a = 0; a < 6; a++) {
            const size_t i = 6 * (size_t)c + a;
            w->xc[i] = cams[i] + w->dc[i];
            if (w->free_idx[c] >= 0) { const double d = w->xc[i] - cams[i]; sn += d * d; xn += cams[i] * cams[i]; }
        }
    for (size_t i = 0; i < 3 * (size_t)w->n_pt; i++) {
        /* bounds on every point coordinate: Ceres projects x + delta onto the box */
        w->xp[i] = clampd(pts[i] + w->dp[i], w->opt->lower_bound, w->opt->upper_bound);
        const double d = w->xp[i] - pts[i];
        sn += d * d; xn += pts[i] * pts[i];
    }
    *step_norm = sqrt(sn); *x_norm = sqrt(xn);
}

static double gradient_max_norm(const ws_t* w)
{
    double m = 0.0;
    for (size_t i = 0; i < 6 * (size_t)w->n_free; i++) if (fabs(w->gc[i]) > m) m = fabs(w->gc[i]);
    for (size_t i = 0; i < 3 * (size_t)w->n_pt; i++) if (fabs(w->gp[i]) > m) m = fabs(w->gp[i]);
    return m;
}

static double gradient_dot_step(const ws_t* w)
{
    double s = 0.0;
    for (uint32_t c = 0; c < w->n_cam; c++) {
        const int32_t f = w->free_idx[c];
        if (f < 0) continue;
        for (int a = 0; a < 6; a++) s += w->gc[6 * (size_t)f + a] * w->dc[6 * (size_t)c + a];
    }
    for (size_t i = 0; i < 3 * (size_t)w->n_pt; i++) s += w->gp[i] * w->dp[i];
    return s;
}

static void set_threads(const oracle_ba_options* opt)
{
#ifdef _OPENMP
    omp_set_num_threads(opt->num_threads > 0 ? opt->num_threads : 1);
#else
    (void)opt;
#endif
}

int oracle_ba_step(uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                   const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                   const double* cams, const double* pts, const double* pl, const double* pr,
                   const uint8_t* cam_fixed, const oracle_ba_options* opt, double radius,
                   double* s_dense, double* rhs, double* dc, double* dp, double* scalars)
{
    ws_t w;
    set_threads(opt);
    if (ws_init(&w, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_uv, pl, pr, cam_fixed, opt) != 0) return -1;
    double cost = oracle_ba_linearize(n_obs, obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, cam_fixed, opt->huber_delta, w.r, w.jc, w.jp);
    build_normal(&w);
    compute_scale(&w);
    compute_envelope(&w);
    int rc = solve_step(&w, radius, s_dense, rhs);
    if (rc == 0) {
        double sn, xn;
        make_candidate(&w, cams, pts, &sn, &xn);
        if (dc) memcpy(dc, w.dc, sizeof(double) * 6 * (size_t)n_cam);
        if (dp) memcpy(dp, w.dp, sizeof(double) * 3 * (size_t)n_pt);
        if (scalars) {
            scalars[0] = cost;
            scalars[1] = model_cost_change(&w);
            scalars[2] = oracle_ba_cost(n_obs, obs_cam, obs_pt, obs_uv, w.xc, w.xp, pl, pr, opt->huber_delta);
            scalars[3] = sn;
        }
    }
    ws_free(&w);
    return rc;
}

/* one rank's pre-reduction payload: its points' Schur contribution plus its own observations' camera blocks */
static int shard_partial(ws_t* w, uint32_t p0, uint32_t p1, double* Sr, double* rr, double* diag)
{
    const size_t n6 = (size_t)w->n_free * 6;
    memset(Sr, 0, sizeof(double) * n6 * n6);
    memset(rr, 0, sizeof(double) * n6);
    if (diag) memset(diag, 0, sizeof(double) * n6);
    int rc = schur_accumulate(w, p0, p1, Sr, rr, NULL);
    for (uint32_t p = p0; p < p1; p++)
        for (uint32_t q = w->pt_start[p]; q < w->pt_start[p + 1]; q++) {
            const uint32_t k = w->pt_obs[q];
            const int32_t f = w->free_idx[w->obs_cam[k]];
            if (f < 0) continue;
            const double* jc = w->jc + 24 * (size_t)k; const double* r = w->r + 4 * (size_t)k;
            for (int i = 0; i < 4; i++)
                for (int a = 0; a < 6; a++) {
                    rr[6 * (size_t)f + a] -= jc[i * 6 + a] * r[i];
                    if (diag) diag[6 * (size_t)f + a] += jc[i * 6 + a] * jc[i * 6 + a];
                    for (int b = a; b < 6; b++) Sr[(6 * (size_t)f + a) * n6 + 6 * (size_t)f + b] += jc[i * 6 + a] * jc[i * 6 + b];
                }
        }
    return rc;
}

int oracle_ba_shard_system(uint32_t rank, uint32_t n_rank, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                           const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                           const double* cams, const double* pts, const double* pl, const double* pr,
                           const uint8_t* cam_fixed, const oracle_ba_options* opt, double radius,
                           double* s_partial, double* rhs_partial, double* diag_partial)
{
    ws_t w;
    set_threads(opt);
    if (n_rank == 0 || rank >= n_rank) return -1;
    if (ws_init(&w, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_uv, pl, pr, cam_fixed, opt) != 0) return -1;
    oracle_ba_linearize(n_obs, obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, cam_fixed, opt->huber_delta, w.r, w.jc, w.jp);
    build_normal(&w);
    compute_scale(&w);    /* point scales are rank-local (all observations of a point live on one rank) */
    compute_damping(&w, radius);
    const uint32_t p0 = (uint32_t)(((uint64_t)rank * n_pt) / n_rank), p1 = (uint32_t)(((uint64_t)(rank + 1) * n_pt) / n_rank);
    int rc = shard_partial(&w, p0, p1, s_partial, rhs_partial, diag_partial);
    const size_t n6 = (size_t)w.n_free * 6;
    for (size_t i = 0; i < n6; i++) for (size_t j = 0; j < i; j++) s_partial[i * n6 + j] = s_partial[j * n6 + i];
    ws_free(&w);
    return rc;
}

int oracle_ba_step_sharded(uint32_t n_rank, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                           const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                           const double* cams, const double* pts, const double* pl, const double* pr,
                           const uint8_t* cam_fixed, const oracle_ba_options* opt, double radius,
                           double* s_dense, double* rhs)
{
    ws_t w;
    set_threads(opt);
    if (n_rank == 0) return -1;
    if (ws_init(&w, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_uv, pl, pr, cam_fixed, opt) != 0) return -1;
    oracle_ba_linearize(n_obs, obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, cam_fixed, opt->huber_delta, w.r, w.jc, w.jp);
    build_normal(&w);   /* B, g_c here are the all-rank sums; per-rank partials are formed below */
    compute_scale(&w);
    compute_damping(&w, radius);
    const size_t n6 = (size_t)w.n_free * 6;
    double* Ssum = (double*)xcalloc(n6 * n6, sizeof(double));
    double* rsum = (double*)xcalloc(n6, sizeof(double));
    double* Sr = (double*)xcalloc(n6 * n6, sizeof(double));
    double* rr = (double*)xcalloc(n6, sizeof(double));
    int rc = 0;
    for (uint32_t rank = 0; rank < n_rank && rc == 0; rank++) {
        const uint32_t p0 = (uint32_t)(((uint64_t)rank * n_pt) / n_rank), p1 = (uint32_t)(((uint64_t)(rank + 1) * n_pt) / n_rank);
        rc = shard_partial(&w, p0, p1, Sr, rr, NULL);
        for (size_t i = 0; i < n6 * n6; i++) Ssum[i] += Sr[i];   /* the all-reduce */
        for (size_t i = 0; i < n6; i++) rsum[i] += rr[i];
    }
    if (rc == 0) {
        for (uint32_t f = 0; f < w.n_free; f++)
            for (int a = 0; a < 6; a++) Ssum[(6 * (size_t)f + a) * n6 + 6 * (size_t)f + a] += w.lc[6 * (size_t)f + a];
        for (size_t i = 0; i < n6; i++) for (size_t j = 0; j < i; j++) Ssum[i * n6 + j] = Ssum[j * n6 + i];
        if (s_dense) memcpy(s_dense, Ssum, sizeof(double) * n6 * n6);
        if (rhs) memcpy(rhs, rsum, sizeof(double) * n6);
    }
    free(Ssum); free(rsum); free(Sr); free(rr);
    ws_free(&w);
    return rc;
}

/* ---- Ceres' line search on bounded problems ---------------------------------------------------------------------
 * /root/reference/src/bundle_adjuster.cpp:104-108 bounds every point coordinate, which makes the problem "constrained":
 * TrustRegionMinimizer::Minimize then calls DoLineSearch(x, gradient, cost, &delta) on every valid step BEFORE the
 * candidate is evaluated - an ARMIJO search (ceres line_search.cc, ArmijoLineSearch::DoSearch) along the projected step
 * with the defaults the reference leaves alone: sufficient decrease 1e-4, CUBIC interpolation (so the gradient is
 * evaluated at every trial), contraction limits [1e-3, 0.6] per step, min step size 1e-9, at most 20 iterations.  On
 * success delta is scaled by the step size found; the model cost change stays that of the full step.
 * Restated from public sources - PARITY UNPINNED like the rest of this file.  Not restated to the
 * bit: Ceres solves the interpolation conditions with Eigen's FullPivLU and finds the roots of a quartic derivative
 * through the eigenvalues of a balanced companion matrix; here partial-pivoted elimination and Aberth iteration give the
 * same numbers to rounding. */
typedef struct { double x, value, gradient; int value_ok, gradient_ok; } ls_sample;

static double poly_eval(const double* c, int deg, double x)   /* c[0] x^deg + ... + c[deg] */
{
    double v = 0.0;
    for (int i = 0; i <= deg; i++) v = v * x + c[i];
    return v;
}

/* coefficients (highest power first) of the polynomial through the samples' values and gradients; returns the degree */
static int find_interpolating_polynomial(const ls_sample* s, int n, double* coef)
{
    int nc = 0;
    for (int i = 0; i < n; i++) nc += s[i].value_ok + s[i].gradient_ok;
    const int deg = nc - 1;
    double a[6][7];
    int row = 0;
    for (int i = 0; i < n; i++) {
        if (s[i].value_ok) {
            for (int j = 0; j <= deg; j++) a[row][j] = pow(s[i].x, deg - j);
            a[row][nc] = s[i].value; row++;
        }
        if (s[i].gradient_ok) {
            for (int j = 0; j < deg; j++) a[row][j] = (deg - j) * pow(s[i].x, deg - j - 1);
            a[row][deg] = 0.0;
            a[row][nc] = s[i].gradient; row++;
        }
    }
    for (int k = 0; k < nc; k++) {   /* Gaussian elimination, partial pivoting */
        int piv = k;
        for (int r = k + 1; r < nc; r++) if (fabs(a[r][k]) > fabs(a[piv][k])) piv = r;
        if (piv != k) for (int j = 0; j <= nc; j++) { const double t = a[k][j]; a[k][j] = a[piv][j]; a[piv][j] = t; }
        if (a[k][k] == 0.0) continue;
        for (int r = k + 1; r < nc; r++) {
            const double f = a[r][k] / a[k][k];
            for (int j = k; j <= nc; j++) a[r][j] -= f * a[k][j];
        }
    }
    for (int k = nc - 1; k >= 0; k--) {
        double v = a[k][nc];
        for (int j = k + 1; j < nc; j++) v -= a[k][j] * coef[j];
        coef[k] = a[k][k] != 0.0 ? v / a[k][k] : 0.0;
    }
    return deg;
}

/* real parts of the roots of c[0] x^deg + ... + c[deg] (deg <= 4); returns how many */
static int polynomial_root_real_parts(const double* c_in, int deg, double* re)
{
    while (deg > 0 && c_in[0] == 0.0) { c_in++; deg--; }   /* RemoveLeadingZeros */
    if (deg <= 0) return 0;
    if (deg == 1) { re[0] = -c_in[1] / c_in[0]; return 1; }
    if (deg == 2) {
        const double a = c_in[0], b = c_in[1], c = c_in[2], D = b * b - 4 * a * c, sD = sqrt(fabs(D));
        if (D >= 0) {
            if (b >= 0) { re[0] = (-b - sD) / (2.0 * a); re[1] = (2.0 * c) / (-b - sD); }
            else { re[0] = (2.0 * c) / (-b + sD); re[1] = (-b + sD) / (2.0 * a); }
        } else { re[0] = re[1] = -b / (2.0 * a); }
        return 2;
    }
    /* Aberth-Ehrlich on the monic polynomial, complex arithmetic by hand */
    double m[5], zr[4], zi[4];
    for (int i = 0; i <= deg; i++) m[i] = c_in[i] / c_in[0];
    double rad = 0.0;
    for (int i = 1; i <= deg; i++) { const double t = pow(fabs(m[i]), 1.0 / i); if (t > rad) rad = t; }
    if (rad == 0.0) { for (int i = 0; i < deg; i++) re[i] = 0.0; return deg; }
    for (int i = 0; i < deg; i++) { const double th = 2.0 * 3.14159265358979323846 * i / deg + 0.4; zr[i] = rad * cos(th); zi[i] = rad * sin(th); }
    for (int it = 0; it < 200; it++) {
        double worst = 0.0;
        for (int i = 0; i < deg; i++) {
            double pr = 1.0, pi = 0.0, dr = 0.0, di = 0.0;   /* p(z) and p'(z) by Horner */
            for (int k = 1; k <= deg; k++) {
                const double ndr = dr * zr[i] - di * zi[i] + pr, ndi = dr * zi[i] + di * zr[i] + pi;
                const double npr = pr * zr[i] - pi * zi[i] + m[k], npi = pr * zi[i] + pi * zr[i];
                dr = ndr; di = ndi; pr = npr; pi = npi;
            }
            /* w = p/p' ; z -= w / (1 - w * sum_{j != i} 1/(z_i - z_j)) */
            const double dd = dr * dr + di * di;
            if (dd == 0.0) continue;
            const double wr = (pr * dr + pi * di) / dd, wi = (pi * dr - pr * di) / dd;
            double sr = 0.0, si = 0.0;
            for (int j = 0; j < deg; j++) if (j != i) {
                const double ar = zr[i] - zr[j], ai = zi[i] - zi[j], ad = ar * ar + ai * ai;
                if (ad > 0.0) { sr += ar / ad; si -= ai / ad; }
            }
            const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr), qd = qr * qr + qi * qi;
            const double cr = qd > 0.0 ? (wr * qr + wi * qi) / qd : wr, ci = qd > 0.0 ? (wi * qr - wr * qi) / qd : wi;
            zr[i] -= cr; zi[i] -= ci;
            const double mag = sqrt(cr * cr + ci * ci);
            if (mag > worst) worst = mag;
        }
        if (worst <= 1e-15 * rad) break;
    }
    for (int i = 0; i < deg; i++) re[i] = zr[i];
    return deg;
}

/* ceres MinimizePolynomial: the midpoint, the two ends and the derivative's roots inside [x_min, x_max] */
static double minimize_polynomial(const double* c, int deg, double x_min, double x_max)
{
    double best_x = 0.5 * (x_min + x_max), best = poly_eval(c, deg, best_x);
    const double v0 = poly_eval(c, deg, x_min);
    if (v0 < best) { best = v0; best_x = x_min; }
    const double v1 = poly_eval(c, deg, x_max);
    if (v1 < best) { best = v1; best_x = x_max; }
    if (deg <= 1) return best_x;
    double d[6], re[4];
    for (int i = 0; i < deg; i++) d[i] = (deg - i) * c[i];
    const int nr = polynomial_root_real_parts(d, deg - 1, re);
    for (int i = 0; i < nr; i++) {
        if (re[i] < x_min || re[i] > x_max) continue;
        const double v = poly_eval(c, deg, re[i]);
        if (v < best) { best = v; best_x = re[i]; }
    }
    return best_x;
}

/* LineSearch::InterpolatingPolynomialMinimizingStepSize with CUBIC interpolation */
static double ls_next_step(const ls_sample* lower, const ls_sample* previous, const ls_sample* current, double min_step, double max_step)
{
    if (!current->value_ok) return fmin(fmax(current->x * 0.5, min_step), max_step);
    ls_sample s[3];
    int n = 0;
    s[n++] = *lower;
    s[n++] = *current;
    if (previous->value_ok) s[n++] = *previous;
    double coef[6];
    const int deg = find_interpolating_polynomial(s, n, coef);
    return minimize_polynomial(coef, deg, min_step, max_step);
}

/* LineSearchFunction::Evaluate at step size a: x+ = Plus(x, a delta) (points projected onto the box), cost and
 * direction . gradient there */
static void ls_evaluate(ws_t* w, const double* cams, const double* pts, double a, ls_sample* out, double* r, double* jc, double* jp)
{
    out->x = a; out->value_ok = 0; out->gradient_ok = 0;
    for (size_t i = 0; i < 6 * (size_t)w->n_cam; i++) w->xc[i] = cams[i] + a * w->dc[i];
    for (size_t i = 0; i < 3 * (size_t)w->n_pt; i++) w->xp[i] = clampd(pts[i] + a * w->dp[i], w->opt->lower_bound, w->opt->upper_bound);
    out->value = oracle_ba_linearize(w->n_obs, w->obs_cam, w->obs_pt, w->uv, w->xc, w->xp, w->pl, w->pr, w->fixed, w->opt->huber_delta,
                                     r, jc, jp);
    if (!isfinite(out->value)) return;
    out->value_ok = 1;
    double g = 0.0;
#pragma omp parallel for reduction(+ : g) schedule(static)
    for (int64_t k = 0; k < (int64_t)w->n_obs; k++) {
        const double* d = w->dc + 6 * (size_t)w->obs_cam[k]; const double* e = w->dp + 3 * (size_t)w->obs_pt[k];
        for (int i = 0; i < 4; i++) {
            double m = jp[12 * (size_t)k + i * 3] * e[0] + jp[12 * (size_t)k + i * 3 + 1] * e[1] + jp[12 * (size_t)k + i * 3 + 2] * e[2];
            for (int q = 0; q < 6; q++) m += jc[24 * (size_t)k + i * 6 + q] * d[q];
            g += m * r[4 * (size_t)k + i];
        }
    }
    out->gradient = g;
    if (isfinite(g)) out->gradient_ok = 1;
}

/* ArmijoLineSearch::DoSearch from step size 1; returns the step size to scale delta by (1 when the search fails or the
 * full step already satisfies the sufficient-decrease condition) and counts its iterations */
static double armijo_line_search(ws_t* w, const double* cams, const double* pts, double x_cost, double g_dot_delta, double full_cost,
                                 int* n_iterations)
{
    const double suff = 1e-4, max_contraction = 1e-3, min_contraction = 0.6, min_step_size = 1e-9;
    const int max_iter = 20;
    if (full_cost <= x_cost + suff * g_dot_delta * 1.0) return 1.0;    /* Evaluate(1.0) already satisfies Armijo */
    double dmax = 0.0;   /* DirectionInfinityNorm */
    for (size_t i = 0; i < 6 * (size_t)w->n_cam; i++) if (fabs(w->dc[i]) > dmax) dmax = fabs(w->dc[i]);
    for (size_t i = 0; i < 3 * (size_t)w->n_pt; i++) if (fabs(w->dp[i]) > dmax) dmax = fabs(w->dp[i]);
    double* r = (double*)malloc(sizeof(double) * 4 * (size_t)w->n_obs);
    double* jc = (double*)malloc(sizeof(double) * 24 * (size_t)w->n_obs);
    double* jp = (double*)malloc(sizeof(double) * 12 * (size_t)w->n_obs);
    if (!r || !jc || !jp) { free(r); free(jc); free(jp); return 1.0; }
    ls_sample initial = {0.0, x_cost, g_dot_delta, 1, 1}, previous = {0.0, 0.0, 0.0, 0, 0}, current;
    ls_evaluate(w, cams, pts, 1.0, &current, r, jc, jp);
    double result = 1.0;
    int iters = 0;
    while (!current.value_ok || current.value > x_cost + suff * g_dot_delta * current.x) {
        iters++;
        if (iters >= max_iter) { result = 1.0; goto done; }                       /* search failed: delta untouched */
        const double step = ls_next_step(&initial, &previous, &current, max_contraction * current.x, min_contraction * current.x);
        if (step * dmax < min_step_size) { result = 1.0; goto done; }
        previous = current;
        ls_evaluate(w, cams, pts, step, &current, r, jc, jp);
    }
    result = current.x;
done:
    *n_iterations += iters;
    free(r); free(jc); free(jp);
    return result;
}

int oracle_ba_solve(uint32_t n_cam, uint32_t n_pt, uint32_t n_obs,
                    const uint32_t* obs_cam, const uint32_t* obs_pt, const float* obs_uv,
                    double* cams, double* pts, const double* pl, const double* pr,
                    const uint8_t* cam_fixed, const oracle_ba_options* opt,
                    oracle_ba_summary* summary, oracle_ba_iteration* log)
{
    ws_t w;
    set_threads(opt);
    const double t_setup = now_sec();
    if (ws_init(&w, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_uv, pl, pr, cam_fixed, opt) != 0) return -1;
    compute_envelope(&w);
    const double t0 = now_sec();

    double radius = opt->initial_radius, decrease_factor = 2.0;
    double x_cost = oracle_ba_linearize(n_obs, obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, cam_fixed, opt->huber_delta, w.r, w.jc, w.jp);
    build_normal(&w);
    compute_scale(&w);
    double gmax = gradient_max_norm(&w);
    int iterations = 0, accepted = 0, invalid_run = 0, armijo = 0, term = ORACLE_TERM_MAX_ITERATIONS, rc = 0;
    const double initial_cost = x_cost;
    if (log) { memset(&log[0], 0, sizeof log[0]); log[0].cost = x_cost; log[0].radius = radius; log[0].gradient_max_norm = gmax; log[0].accepted = 1; log[0].valid = 1; }

    if (opt->check_termination && gmax <= opt->gradient_tolerance) term = ORACLE_TERM_GRADIENT_TOLERANCE;
    else while (1) {
        if (iterations >= opt->max_iterations) { term = ORACLE_TERM_MAX_ITERATIONS; break; }
        if (opt->check_termination && radius < opt->min_radius) { term = ORACLE_TERM_MIN_RADIUS; break; }
        iterations++;
        oracle_ba_iteration it; memset(&it, 0, sizeof it);
        it.radius = radius; it.cost = x_cost; it.gradient_max_norm = gmax;
        int lin = solve_step(&w, radius, NULL, NULL);
        double mcc = lin == 0 ? model_cost_change(&w) : 0.0;
        it.model_cost_change = mcc;
        if (lin != 0 || !(mcc > 0.0)) {
            /* invalid step (TrustRegionMinimizer::HandleInvalidStep) */
            it.valid = 0;
            if (log) log[iterations] = it;
            if (opt->check_termination && ++invalid_run >= 5) { term = ORACLE_TERM_INVALID_STEPS; break; }
            /* LevenbergMarquardtStrategy::StepIsInvalid: radius *= 0.5, the rejected-step factor is left alone */
            radius *= 0.5;
            continue;
        }
        invalid_run = 0; it.valid = 1;
        double step_norm, x_norm;
        make_candidate(&w, cams, pts, &step_norm, &x_norm);
        double cand = oracle_ba_cost(n_obs, obs_cam, obs_pt, obs_uv, w.xc, w.xp, pl, pr, opt->huber_delta);
        /* bounded problem: DoLineSearch along the projected step before the candidate is evaluated; on success delta is
           scaled by the step size found (the model cost change stays the full step's) */
        {
            const double a = armijo_line_search(&w, cams, pts, x_cost, gradient_dot_step(&w), cand, &armijo);
            if (a != 1.0) {
                for (size_t i = 0; i < 6 * (size_t)n_cam; i++) w.dc[i] *= a;
                for (size_t i = 0; i < 3 * (size_t)n_pt; i++) w.dp[i] *= a;
                make_candidate(&w, cams, pts, &step_norm, &x_norm);
                cand = oracle_ba_cost(n_obs, obs_cam, obs_pt, obs_uv, w.xc, w.xp, pl, pr, opt->huber_delta);
            }
        }
        it.candidate_cost = cand; it.step_norm = step_norm;
        if (opt->check_termination) {
            if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
                if (log) log[iterations] = it;
                term = ORACLE_TERM_PARAMETER_TOLERANCE; break;
            }
            if (fabs(x_cost - cand) <= opt->function_tolerance * x_cost) {
                if (log) log[iterations] = it;
                term = ORACLE_TERM_FUNCTION_TOLERANCE; break;
            }
        }
        const double rel = (x_cost - cand) / mcc;
        it.relative_decrease = rel;
        if (rel > opt->min_relative_decrease) {
            memcpy(cams, w.xc, sizeof(double) * 6 * (size_t)n_cam);
            memcpy(pts, w.xp, sizeof(double) * 3 * (size_t)n_pt);
            x_cost = oracle_ba_linearize(n_obs, obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, cam_fixed, opt->huber_delta, w.r, w.jc, w.jp);
            build_normal(&w);
            gmax = gradient_max_norm(&w);
            double f = 1.0 - pow(2.0 * rel - 1.0, 3.0);
            if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            radius = radius / f;
            if (radius > opt->max_radius) radius = opt->max_radius;
            decrease_factor = 2.0;
            accepted++; it.accepted = 1; it.cost = x_cost; it.gradient_max_norm = gmax;
            if (log) log[iterations] = it;
            if (opt->check_termination && gmax <= opt->gradient_tolerance) { term = ORACLE_TERM_GRADIENT_TOLERANCE; break; }
        } else {
            radius /= decrease_factor; decrease_factor *= 2.0;
            if (log) log[iterations] = it;
        }
    }
    if (summary) {
        summary->initial_cost = initial_cost; summary->final_cost = x_cost;
        summary->iterations = iterations; summary->accepted = accepted; summary->termination = term;
        summary->line_search_steps = armijo;
        summary->solve_seconds = now_sec() - t0; summary->setup_seconds = t0 - t_setup;
    }
    ws_free(&w);
    return rc;
}
