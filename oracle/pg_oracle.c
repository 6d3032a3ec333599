/*
 * pg_oracle.c - CPU restatement of the g2o solve inside PoseGraphOptimizer::Optimize
 * (/root/reference/src/pose_graph_optimizer.cpp:61-69).  TEST INFRASTRUCTURE, PARITY UNPINNED - see pg_oracle.h.
 *
 * Restated pieces:
 *   VertexSE3            estimate [t, q.xyz, q.w]; oplus: X <- X * fromVectorMQT(d), q.w = sqrt(1 - |d.q|^2)
 *   EdgeSE3              e = toVectorMQT(Z^-1 * Xi^-1 * Xj), exact derivatives wrt both 6-dof increments
 *   information          one shared 6x6, diag(.01,.01,.01,1,1,1) in the reference (:23-26)
 *   RobustKernelHuber    rho(chi2), weight rho' on Omega, no second-order term
 *   Levenberg            lambda0 = tau * max diag(H); trial loop with push/pop; rho = dchi / (x.(lambda x + b) + 1e-3)
 *   linear solver        the reference's is a sparse Cholesky (LinearSolverEigen); here dense Cholesky for small
 *                        systems and block-Jacobi PCG (tolerance 1e-12) above 1500 unknowns
 */
#define _POSIX_C_SOURCE 200809L
#include "pg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void oracle_pg_options_default(oracle_pg_options* o)
{
    o->max_iterations = 10;
    o->max_trials = 10;
    o->huber_delta = 1.0;
    o->initial_lambda_scale = 1e-5;
    o->num_threads = 1;
    o->reserved = 0;
}

/* ---- quaternion / isometry helpers: q = (x, y, z, w) --------------------------------------------------- */

static void q_mul(const double* a, const double* b, double* c)
{
    const double x = a[3] * b[0] + b[3] * a[0] + a[1] * b[2] - a[2] * b[1];
    const double y = a[3] * b[1] + b[3] * a[1] + a[2] * b[0] - a[0] * b[2];
    const double z = a[3] * b[2] + b[3] * a[2] + a[0] * b[1] - a[1] * b[0];
    const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    c[0] = x; c[1] = y; c[2] = z; c[3] = w;
}

static void q_conj(const double* a, double* c) { c[0] = -a[0]; c[1] = -a[1]; c[2] = -a[2]; c[3] = a[3]; }

static void q_normalize(double* q)
{
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}

static void q_to_rot(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

static void q_rotate(const double* q, const double* v, double* out)
{
    double R[9];
    q_to_rot(q, R);
    for (int i = 0; i < 3; i++) out[i] = R[i * 3] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2];
}

static void q_rotate_inv(const double* q, const double* v, double* out)
{
    double R[9];
    q_to_rot(q, R);
    for (int i = 0; i < 3; i++) out[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}

/* e = toVectorMQT(Z^-1 Xi^-1 Xj) and its derivatives wrt the increments of Xi and Xj */
void oracle_pg_edge(const double* xi, const double* xj, const double* z, double* e, double* ji, double* jj)
{
    double qi[4] = {xi[3], xi[4], xi[5], xi[6]}, qj[4] = {xj[3], xj[4], xj[5], xj[6]}, qz[4] = {z[3], z[4], z[5], z[6]};
    q_normalize(qi); q_normalize(qj); q_normalize(qz);
    double qic[4], qzc[4], qa[4], qe[4];
    q_conj(qi, qic); q_conj(qz, qzc);
    q_mul(qic, qj, qa);          /* rotation of A = Xi^-1 Xj */
    q_mul(qzc, qa, qe);          /* rotation of E = Z^-1 A   */
    double d[3] = {xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2]}, ta[3], tmp[3], te[3];
    q_rotate_inv(qi, d, ta);     /* translation of A */
    tmp[0] = ta[0] - z[0]; tmp[1] = ta[1] - z[1]; tmp[2] = ta[2] - z[2];
    q_rotate_inv(qz, tmp, te);
    const double sgn = qe[3] < 0.0 ? -1.0 : 1.0;
    e[0] = te[0]; e[1] = te[1]; e[2] = te[2];
    e[3] = sgn * qe[0]; e[4] = sgn * qe[1]; e[5] = sgn * qe[2];
    if (!ji && !jj) return;

    double Rz[9], Re[9];
    q_to_rot(qz, Rz);
    q_to_rot(qe, Re);
    if (jj) {
        memset(jj, 0, 36 * sizeof(double));
        /* d te / d dt_j = Re ; d qe / d dq_j: qe * (dq, 1) -> w I + [v]x, v = vec(qe) */
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) jj[r * 6 + c] = Re[r * 3 + c];
        const double w = sgn * qe[3], v0 = sgn * qe[0], v1 = sgn * qe[1], v2 = sgn * qe[2];
        jj[3 * 6 + 3] = w;   jj[3 * 6 + 4] = -v2; jj[3 * 6 + 5] = v1;
        jj[4 * 6 + 3] = v2;  jj[4 * 6 + 4] = w;   jj[4 * 6 + 5] = -v0;
        jj[5 * 6 + 3] = -v1; jj[5 * 6 + 4] = v0;  jj[5 * 6 + 5] = w;
    }
    if (ji) {
        memset(ji, 0, 36 * sizeof(double));
        /* d te / d dt_i = -Rz^T ; d te / d dq_i = 2 Rz^T [ta]x */
        const double tx[9] = {0, -ta[2], ta[1], ta[2], 0, -ta[0], -ta[1], ta[0], 0};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                ji[r * 6 + c] = -Rz[c * 3 + r];
                ji[r * 6 + 3 + c] = 2.0 * (Rz[0 * 3 + r] * tx[0 * 3 + c] + Rz[1 * 3 + r] * tx[1 * 3 + c] + Rz[2 * 3 + r] * tx[2 * 3 + c]);
            }
        /* d qe / d dq_i: qe' = conj(qz) * (-dq, 1) * qa  ->  column k = -vec(conj(qz) * e_k * qa) */
        for (int k = 0; k < 3; k++) {
            double ek[4] = {0, 0, 0, 0}, t1[4], t2[4];
            ek[k] = 1.0;
            q_mul(qzc, ek, t1);
            q_mul(t1, qa, t2);
            for (int r = 0; r < 3; r++) ji[(3 + r) * 6 + 3 + k] = -sgn * t2[r];
        }
    }
}

static void huber(double e2, double delta, double* rho0, double* rho1)
{
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.0; }
    else { const double s = sqrt(e2); *rho0 = 2.0 * s * delta - dsqr; *rho1 = delta / s; }
}

static double quad6(const double* info, const double* e)
{
    double s = 0.0;
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) s += e[a] * info[a * 6 + b] * e[b];
    return s;
}

double oracle_pg_chi2(uint32_t n_vertex, uint32_t n_edge, const double* est, const uint32_t* ef, const uint32_t* et,
                      const double* meas, const double* info, double delta, double* edge_chi2)
{
    (void)n_vertex;
    double sum = 0.0;
    for (uint32_t k = 0; k < n_edge; k++) {
        double e[6], r0, r1;
        oracle_pg_edge(est + 7 * (size_t)ef[k], est + 7 * (size_t)et[k], meas + 7 * (size_t)k, e, NULL, NULL);
        const double c = quad6(info, e);
        if (edge_chi2) edge_chi2[k] = c;
        huber(c, delta, &r0, &r1);
        sum += r0;
    }
    return sum;
}

/* ---- per-edge linearisation kept matrix-free ----------------------------------------------------------- */

typedef struct pg_ws {
    uint32_t n_vertex, n_edge, n_free;
    const uint32_t* ef; const uint32_t* et; const double* meas; const double* info; double delta;
    int32_t* free_idx;
    double* ji; double* jj;   /* n_edge * 36 */
    double* wom;              /* n_edge * 36 : rho' * Omega */
    double* D;                /* n_free * 36 : diagonal blocks of H */
    double* b;                /* n_free * 6 */
} pg_ws;

static double pg_linearize(pg_ws* w, const double* est)
{
    memset(w->D, 0, sizeof(double) * 36 * (size_t)w->n_free);
    memset(w->b, 0, sizeof(double) * 6 * (size_t)w->n_free);
    double chi = 0.0;
    for (uint32_t k = 0; k < w->n_edge; k++) {
        double e[6], r0, r1;
        double* Ji = w->ji + 36 * (size_t)k; double* Jj = w->jj + 36 * (size_t)k; double* W = w->wom + 36 * (size_t)k;
        oracle_pg_edge(est + 7 * (size_t)w->ef[k], est + 7 * (size_t)w->et[k], w->meas + 7 * (size_t)k, e, Ji, Jj);
        huber(quad6(w->info, e), w->delta, &r0, &r1);
        chi += r0;
        for (int a = 0; a < 36; a++) W[a] = r1 * w->info[a];
        double We[6];
        for (int a = 0; a < 6; a++) { We[a] = 0.0; for (int c = 0; c < 6; c++) We[a] += W[a * 6 + c] * e[c]; }
        const int32_t fi = w->free_idx[w->ef[k]], fj = w->free_idx[w->et[k]];
        const double* J[2] = {Ji, Jj};
        const int32_t f[2] = {fi, fj};
        for (int s = 0; s < 2; s++) {
            if (f[s] < 0) continue;
            double WJ[36];
            for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { double t = 0; for (int m = 0; m < 6; m++) t += W[a * 6 + m] * J[s][m * 6 + c]; WJ[a * 6 + c] = t; }
            double* Dk = w->D + 36 * (size_t)f[s];
            for (int a = 0; a < 6; a++) {
                double t = 0; for (int m = 0; m < 6; m++) t += J[s][m * 6 + a] * We[m];
                w->b[6 * (size_t)f[s] + a] -= t;
                for (int c = 0; c < 6; c++) { double u = 0; for (int m = 0; m < 6; m++) u += J[s][m * 6 + a] * WJ[m * 6 + c]; Dk[a * 6 + c] += u; }
            }
        }
    }
    return chi;
}

/* y = H x (no damping) */
static void pg_matvec(const pg_ws* w, const double* x, double* y)
{
    memset(y, 0, sizeof(double) * 6 * (size_t)w->n_free);
    for (uint32_t k = 0; k < w->n_edge; k++) {
        const int32_t fi = w->free_idx[w->ef[k]], fj = w->free_idx[w->et[k]];
        const double* Ji = w->ji + 36 * (size_t)k; const double* Jj = w->jj + 36 * (size_t)k; const double* W = w->wom + 36 * (size_t)k;
        double t[6] = {0, 0, 0, 0, 0, 0}, u[6];
        if (fi >= 0) for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) t[a] += Ji[a * 6 + c] * x[6 * (size_t)fi + c];
        if (fj >= 0) for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) t[a] += Jj[a * 6 + c] * x[6 * (size_t)fj + c];
        for (int a = 0; a < 6; a++) { u[a] = 0; for (int c = 0; c < 6; c++) u[a] += W[a * 6 + c] * t[c]; }
        if (fi >= 0) for (int a = 0; a < 6; a++) for (int m = 0; m < 6; m++) y[6 * (size_t)fi + a] += Ji[m * 6 + a] * u[m];
        if (fj >= 0) for (int a = 0; a < 6; a++) for (int m = 0; m < 6; m++) y[6 * (size_t)fj + a] += Jj[m * 6 + a] * u[m];
    }
}

static void pg_dense(const pg_ws* w, double* H)
{
    const size_t n6 = 6 * (size_t)w->n_free;
    memset(H, 0, sizeof(double) * n6 * n6);
    for (uint32_t k = 0; k < w->n_edge; k++) {
        const int32_t f[2] = {w->free_idx[w->ef[k]], w->free_idx[w->et[k]]};
        const double* J[2] = {w->ji + 36 * (size_t)k, w->jj + 36 * (size_t)k};
        const double* W = w->wom + 36 * (size_t)k;
        for (int s = 0; s < 2; s++) {
            if (f[s] < 0) continue;
            double WJ[36];
            for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { double t = 0; for (int m = 0; m < 6; m++) t += W[a * 6 + m] * J[s][m * 6 + c]; WJ[a * 6 + c] = t; }
            for (int r = 0; r < 2; r++) {
                if (f[r] < 0) continue;
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c < 6; c++) {
                        double u = 0; for (int m = 0; m < 6; m++) u += J[r][m * 6 + a] * WJ[m * 6 + c];
                        H[(6 * (size_t)f[r] + a) * n6 + 6 * (size_t)f[s] + c] += u;
                    }
            }
        }
    }
}

static int dense_cholesky_solve(double* A, size_t n, double* b)
{
    for (size_t i = 0; i < n; i++) {
        for (size_t j = 0; j <= i; j++) {
            double s = A[i * n + j];
            for (size_t k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            if (j < i) A[i * n + j] = s / A[j * n + j];
            else { if (!(s > 0.0)) return -1; A[i * n + i] = sqrt(s); }
        }
    }
    for (size_t i = 0; i < n; i++) { double s = b[i]; for (size_t k = 0; k < i; k++) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
    for (size_t i = n; i-- > 0;) { double s = b[i]; for (size_t k = i + 1; k < n; k++) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
    return 0;
}

static int inv6(const double* A, double* out)
{
    double L[36];
    memset(L, 0, sizeof L);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0.0)) return -1; L[i * 6 + i] = sqrt(s); } else L[i * 6 + j] = s / L[j * 6 + j];
        }
    for (int c = 0; c < 6; c++) {
        double y[6];
        for (int i = 0; i < 6; i++) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
        for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * out[k * 6 + c]; out[i * 6 + c] = s / L[i * 6 + i]; }
    }
    return 0;
}

/* (H + lambda I) x = b by block-Jacobi PCG; returns iterations or -1 */
static int pg_pcg(const pg_ws* w, double lambda, double* x, double tol, int max_iter)
{
    const size_t n = 6 * (size_t)w->n_free;
    double* r = (double*)malloc(sizeof(double) * n * 4);
    double* Mi = (double*)malloc(sizeof(double) * 36 * (size_t)w->n_free);
    if (!r || !Mi) { free(r); free(Mi); return -1; }
    double *z = r + n, *p = r + 2 * n, *q = r + 3 * n;
    for (uint32_t f = 0; f < w->n_free; f++) {
        double B[36];
        memcpy(B, w->D + 36 * (size_t)f, sizeof B);
        for (int a = 0; a < 6; a++) B[a * 7] += lambda;
        if (inv6(B, Mi + 36 * (size_t)f) != 0) { free(r); free(Mi); return -1; }
    }
    double bb = 0, rz = 0;
    for (size_t i = 0; i < n; i++) { x[i] = 0; r[i] = w->b[i]; bb += r[i] * r[i]; }
    for (uint32_t f = 0; f < w->n_free; f++)
        for (int a = 0; a < 6; a++) { double s = 0; for (int c = 0; c < 6; c++) s += Mi[36 * (size_t)f + a * 6 + c] * r[6 * (size_t)f + c]; z[6 * (size_t)f + a] = s; }
    for (size_t i = 0; i < n; i++) { p[i] = z[i]; rz += r[i] * z[i]; }
    int it = 0;
    double rr = bb;
    while (it < max_iter && rr > tol * tol * bb) {
        pg_matvec(w, p, q);
        double pq = 0;
        for (size_t i = 0; i < n; i++) { q[i] += lambda * p[i]; pq += p[i] * q[i]; }
        const double alpha = rz / pq;
        rr = 0;
        for (size_t i = 0; i < n; i++) { x[i] += alpha * p[i]; r[i] -= alpha * q[i]; rr += r[i] * r[i]; }
        for (uint32_t f = 0; f < w->n_free; f++)
            for (int a = 0; a < 6; a++) { double s = 0; for (int c = 0; c < 6; c++) s += Mi[36 * (size_t)f + a * 6 + c] * r[6 * (size_t)f + c]; z[6 * (size_t)f + a] = s; }
        double rz2 = 0;
        for (size_t i = 0; i < n; i++) rz2 += r[i] * z[i];
        const double beta = rz2 / rz;
        rz = rz2;
        for (size_t i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
        it++;
    }
    free(r); free(Mi);
    return it;
}

static int pg_ws_init(pg_ws* w, uint32_t n_vertex, uint32_t n_edge, const uint8_t* fixed, const uint32_t* ef, const uint32_t* et,
                      const double* meas, const double* info, double delta)
{
    memset(w, 0, sizeof *w);
    for (uint32_t k = 0; k < n_edge; k++) if (ef[k] >= n_vertex || et[k] >= n_vertex) return -1;
    w->n_vertex = n_vertex; w->n_edge = n_edge; w->ef = ef; w->et = et; w->meas = meas; w->info = info; w->delta = delta;
    w->free_idx = (int32_t*)malloc(sizeof(int32_t) * (n_vertex ? n_vertex : 1));
    uint32_t nf = 0;
    for (uint32_t v = 0; v < n_vertex; v++) w->free_idx[v] = (fixed && fixed[v]) ? -1 : (int32_t)nf++;
    w->n_free = nf;
    w->ji = (double*)malloc(sizeof(double) * 36 * (size_t)(n_edge ? n_edge : 1));
    w->jj = (double*)malloc(sizeof(double) * 36 * (size_t)(n_edge ? n_edge : 1));
    w->wom = (double*)malloc(sizeof(double) * 36 * (size_t)(n_edge ? n_edge : 1));
    w->D = (double*)malloc(sizeof(double) * 36 * (size_t)(nf ? nf : 1));
    w->b = (double*)malloc(sizeof(double) * 6 * (size_t)(nf ? nf : 1));
    return (w->free_idx && w->ji && w->jj && w->wom && w->D && w->b) ? 0 : -2;
}

static void pg_ws_free(pg_ws* w)
{
    free(w->free_idx); free(w->ji); free(w->jj); free(w->wom); free(w->D); free(w->b);
    memset(w, 0, sizeof *w);
}

double oracle_pg_linearize(uint32_t n_vertex, uint32_t n_edge, const double* est, const uint8_t* fixed,
                           const uint32_t* ef, const uint32_t* et, const double* meas, const double* info,
                           double delta, double* h_dense, double* b)
{
    pg_ws w;
    if (pg_ws_init(&w, n_vertex, n_edge, fixed, ef, et, meas, info, delta) != 0) return -1.0;
    const double chi = pg_linearize(&w, est);
    if (h_dense) pg_dense(&w, h_dense);
    if (b) memcpy(b, w.b, sizeof(double) * 6 * (size_t)w.n_free);
    pg_ws_free(&w);
    return chi;
}

/* X <- X * fromVectorMQT(d) for every free vertex (VertexSE3::oplusImpl) */
static void pg_apply(const pg_ws* w, const double* est, const double* x, double* out)
{
    for (uint32_t v = 0; v < w->n_vertex; v++) {
        const double* s = est + 7 * (size_t)v;
        double* o = out + 7 * (size_t)v;
        const int32_t f = w->free_idx[v];
        if (f < 0) { memcpy(o, s, 7 * sizeof(double)); continue; }
        const double* d = x + 6 * (size_t)f;
        double q[4] = {s[3], s[4], s[5], s[6]}, dq[4], t[3], qn[4];
        q_normalize(q);
        const double w2 = 1.0 - (d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
        if (w2 < 0.0) { dq[0] = dq[1] = dq[2] = 0.0; dq[3] = 1.0; }   /* fromCompactQuaternion: identity */
        else { dq[0] = d[3]; dq[1] = d[4]; dq[2] = d[5]; dq[3] = sqrt(w2); }
        q_rotate(q, d, t);
        q_mul(q, dq, qn);
        q_normalize(qn);
        o[0] = s[0] + t[0]; o[1] = s[1] + t[1]; o[2] = s[2] + t[2];
        o[3] = qn[0]; o[4] = qn[1]; o[5] = qn[2]; o[6] = qn[3];
    }
}

int oracle_pg_solve(uint32_t n_vertex, uint32_t n_edge, double* est, const uint8_t* fixed, const uint32_t* ef,
                    const uint32_t* et, const double* meas, const double* info, const oracle_pg_options* opt,
                    oracle_pg_summary* summary, oracle_pg_iteration* log)
{
    pg_ws w;
    const double t_setup = now_sec();
    if (pg_ws_init(&w, n_vertex, n_edge, fixed, ef, et, meas, info, opt->huber_delta) != 0) return -1;
    const size_t n = 6 * (size_t)w.n_free;
    const int use_dense = n <= 1500;
    double* H = use_dense ? (double*)malloc(sizeof(double) * (n ? n * n : 1)) : NULL;
    double* x = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* cand = (double*)malloc(sizeof(double) * 7 * (size_t)(n_vertex ? n_vertex : 1));
    const double t0 = now_sec();
    double lambda = 0.0, ni = 2.0, chi0 = 0.0, chi_final = 0.0;
    int term = ORACLE_PG_TERM_ITERATIONS, it = 0;
    for (it = 0; it < opt->max_iterations; it++) {
        double current = pg_linearize(&w, est);
        if (it == 0) {
            chi0 = current;
            double md = 0.0;
            for (size_t f = 0; f < w.n_free; f++) for (int a = 0; a < 6; a++) md = fmax(md, fabs(w.D[36 * f + a * 7]));
            lambda = opt->initial_lambda_scale * md;
            ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0, accepted = 0;
        do {
            int ok = 1;
            if (n) {
                if (use_dense) {
                    pg_dense(&w, H);
                    for (size_t i = 0; i < n; i++) H[i * n + i] += lambda;
                    memcpy(x, w.b, sizeof(double) * n);
                    ok = dense_cholesky_solve(H, n, x) == 0;
                } else {
                    ok = pg_pcg(&w, lambda, x, 1e-12, 4000) >= 0;
                }
            }
            pg_apply(&w, est, x, cand);
            double temp = oracle_pg_chi2(n_vertex, n_edge, cand, ef, et, meas, info, opt->huber_delta, NULL);
            if (!ok) temp = 1.7976931348623157e308;
            double scale = 0.0;
            for (size_t i = 0; i < n; i++) scale += x[i] * (lambda * x[i] + w.b[i]);
            scale += 1e-3;
            rho = (current - temp) / scale;
            if (rho > 0 && isfinite(temp)) {
                double alpha = 1.0 - pow(2.0 * rho - 1.0, 3.0);
                if (alpha > 2.0 / 3.0) alpha = 2.0 / 3.0;
                const double sf = alpha < 1.0 / 3.0 ? 1.0 / 3.0 : alpha;
                lambda *= sf;
                ni = 2.0;
                current = temp;
                memcpy(est, cand, sizeof(double) * 7 * (size_t)n_vertex);
                accepted = 1;
            } else {
                lambda *= ni;
                ni *= 2.0;
                if (!isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < opt->max_trials);
        chi_final = current;
        if (log) { log[it].chi2 = current; log[it].lambda = lambda; log[it].trials = qmax; log[it].accepted = accepted; }
        if (qmax == opt->max_trials || rho == 0 || !isfinite(lambda)) { term = ORACLE_PG_TERM_TRIALS; it++; break; }
    }
    if (summary) {
        summary->initial_chi2 = chi0; summary->final_chi2 = chi_final; summary->iterations = it; summary->termination = term;
        summary->solve_seconds = now_sec() - t0; summary->setup_seconds = t0 - t_setup;
    }
    free(H); free(x); free(cand);
    pg_ws_free(&w);
    return 0;
}
