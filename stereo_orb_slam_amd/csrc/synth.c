/*
 * synth.c - deterministic synthetic BA / pose-graph workloads (see
 * include/soslam_synth.h for the contract and the reference conventions).
 *
 * Everything random is a pure function of (seed, stream, index): no generator
 * state, so the numpy mirror can evaluate any element out of order.
 */
#include "soslam_synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GOLD 0x9E3779B97F4A7C15ull
#define TWO_PI 6.283185307179586476925286766559

/* stream ids (keep in sync with stereo_orb_slam_amd/synth.py) */
enum {
    ST_TRACK = 1, ST_PX = 2, ST_PY = 3, ST_PZ = 4, ST_NOISE = 5, ST_OSEL = 6,
    ST_OVAL = 7, ST_POSE = 8, ST_DEPTH = 9,
    ST_PG_MEAS = 20, ST_PG_INIT = 21
};

#define MAX_TRACK 64u

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t soslam_synth_u64(uint64_t seed, uint64_t stream, uint64_t index)
{
    uint64_t k = mix64(seed + GOLD * (stream + 1));
    return mix64(k + GOLD * (index + 1));
}

double soslam_synth_uniform(uint64_t seed, uint64_t stream, uint64_t index)
{
    return (double)(soslam_synth_u64(seed, stream, index) >> 11) * (1.0 / 9007199254740992.0);
}

double soslam_synth_normal(uint64_t seed, uint64_t stream, uint64_t index)
{
    double u1 = soslam_synth_uniform(seed, stream, 2 * index);
    double u2 = soslam_synth_uniform(seed, stream, 2 * index + 1);
    return sqrt(-2.0 * log(1.0 - u1)) * cos(TWO_PI * u2);
}

/* ---- small dense helpers (row-major 3x3) -------------------------------- */

static void mat3_mul(const double* a, const double* b, double* c)
{
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            t[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
    memcpy(c, t, sizeof t);
}

static void rot_x(double a, double* r)
{
    double c = cos(a), s = sin(a);
    double m[9] = {1, 0, 0, 0, c, -s, 0, s, c};
    memcpy(r, m, sizeof m);
}
static void rot_y(double a, double* r)
{
    double c = cos(a), s = sin(a);
    double m[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
    memcpy(r, m, sizeof m);
}
static void rot_z(double a, double* r)
{
    double c = cos(a), s = sin(a);
    double m[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
    memcpy(r, m, sizeof m);
}

/* Rodrigues: rotation matrix of the angle-axis vector w. */
static void so3_exp(const double* w, double* r)
{
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double th = sqrt(th2);
    double a, b;
    if (th < 1e-12) { a = 1.0; b = 0.5; }
    else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
    double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double wx2[9];
    mat3_mul(wx, wx, wx2);
    for (int i = 0; i < 9; i++) r[i] = a * wx[i] + b * wx2[i];
    r[0] += 1.0; r[4] += 1.0; r[8] += 1.0;
}

/* Unit quaternion (x y z w) of a rotation matrix, Shepperd's branches. */
static void mat3_to_quat(const double* m, double* q)
{
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        double s = sqrt(t + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (m[7] - m[5]) * s;
        q[1] = (m[2] - m[6]) * s;
        q[2] = (m[3] - m[1]) * s;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
    }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= n;
    if (q[3] < 0.0) for (int a = 0; a < 4; a++) q[a] = -q[a];
}

static double f32r(double v) { return (double)(float)v; }

/* ---- BA workloads -------------------------------------------------------- */

int soslam_synth_ba_config(int config, soslam_synth_ba_params* p)
{
    if (!p) return -1;
    memset(p, 0, sizeof *p);
    p->seed = SOSLAM_SYNTH_SEED;
    p->curvature = 0.002;
    p->pixel_sigma = 0.5;
    p->outlier_frac = 0.05;
    p->outlier_px = 20.0;
    p->pose_rot_sigma = 0.005;
    p->pose_trans_sigma = 0.05;
    p->depth_noise = 0.02;
    switch (config) {
    case 1:
        p->n_cam = 10; p->n_pt = 2000; p->track_mode = SOSLAM_TRACK_GEOMETRIC;
        p->track_len = 8; p->spacing = 0.9; p->curvature = 0.01;  /* mean 8 before the remaining-frames cap => effective mean ~4, ~7.9k obs */
        return 0;
    case 2:
        p->n_cam = 100; p->n_pt = 20000; p->track_mode = SOSLAM_TRACK_FIXED;
        p->track_len = 10; p->spacing = 0.5;
        return 0;
    case 3:
        p->n_cam = 500; p->n_pt = 100000; p->track_mode = SOSLAM_TRACK_FIXED;
        p->track_len = 10; p->spacing = 0.5;
        return 0;
    default:
        return -1;
    }
}

static uint32_t first_cam(const soslam_synth_ba_params* p, uint32_t j)
{
    if (p->track_mode == SOSLAM_TRACK_FIXED) {
        uint32_t span = p->n_cam - p->track_len + 1;
        return (uint32_t)(((uint64_t)j * span) / p->n_pt);
    }
    return (uint32_t)(((uint64_t)j * p->n_cam) / p->n_pt);
}

static uint32_t track_length(const soslam_synth_ba_params* p, uint32_t j, uint32_t c0)
{
    uint32_t len;
    if (p->track_mode == SOSLAM_TRACK_FIXED) {
        len = p->track_len;
    } else {
        double q = 1.0 / (double)p->track_len;
        double u = soslam_synth_uniform(p->seed, ST_TRACK, j);
        len = 1u + (uint32_t)floor(log(1.0 - u) / log(1.0 - q));
    }
    if (len > p->n_cam - c0) len = p->n_cam - c0;
    if (len > MAX_TRACK) len = MAX_TRACK;
    return len;
}

static int ba_params_ok(const soslam_synth_ba_params* p)
{
    if (!p || p->n_cam == 0 || p->n_pt == 0 || p->track_len == 0) return 0;
    if (p->track_mode == SOSLAM_TRACK_FIXED && p->track_len > p->n_cam) return 0;
    if (p->track_mode == SOSLAM_TRACK_GEOMETRIC && p->track_len < 2) return 0;
    if (p->track_len > MAX_TRACK) return 0;
    return 1;
}

int soslam_synth_ba_count(const soslam_synth_ba_params* p, uint32_t* n_obs)
{
    if (!ba_params_ok(p) || !n_obs) return -1;
    uint64_t n = 0;
    for (uint32_t j = 0; j < p->n_pt; j++) n += track_length(p, j, first_cam(p, j));
    if (n > 0xFFFFFFFFull) return -1;
    *n_obs = (uint32_t)n;
    return 0;
}

static void fill_projections(double* pl, double* pr)
{
    const double fx = f32r(718.856), cx = f32r(607.1928), cy = f32r(185.2157);
    const double tx = f32r(-386.1448);
    double l[12] = {fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0};
    double r[12] = {fx, 0, cx, tx, 0, fx, cy, 0, 0, 0, 1, 0};
    if (pl) memcpy(pl, l, sizeof l);
    if (pr) memcpy(pr, r, sizeof r);
}

/* truth camera->world of camera k: rot (row-major 3x3) and position */
static void truth_path(const soslam_synth_ba_params* p, double* rot, double* pos)
{
    double x = 0.0, z = 0.0;
    for (uint32_t k = 0; k < p->n_cam; k++) {
        double s = p->spacing * (double)k;
        double yaw = p->curvature * s;
        double ry[9], rx[9], rz[9], t[9];
        rot_y(yaw, ry);
        rot_x(0.02 * sin(0.1 * s), rx);
        rot_z(0.015 * cos(0.07 * s), rz);
        mat3_mul(ry, rx, t);
        mat3_mul(t, rz, rot + 9 * (size_t)k);
        pos[3 * (size_t)k + 0] = x;
        pos[3 * (size_t)k + 1] = 0.05 * sin(0.05 * s);
        pos[3 * (size_t)k + 2] = z;
        x += p->spacing * sin(yaw);
        z += p->spacing * cos(yaw);
    }
}

static void project(const double* P, const double* pc, double* u, double* v)
{
    double d = P[8] * pc[0] + P[9] * pc[1] + P[10] * pc[2] + P[11];
    *u = (P[0] * pc[0] + P[1] * pc[1] + P[2] * pc[2] + P[3]) / d;
    *v = (P[4] * pc[0] + P[5] * pc[1] + P[6] * pc[2] + P[7]) / d;
}

int soslam_synth_ba_generate(const soslam_synth_ba_params* p,
                             float* poses_wc, float* points,
                             uint32_t* obs_frame, uint32_t* obs_point, float* obs_uv,
                             double* proj_l, double* proj_r,
                             double* true_poses_wc, double* true_points)
{
    if (!ba_params_ok(p)) return -1;
    const uint32_t nc = p->n_cam, np = p->n_pt;
    double Pl[12], Pr[12];
    fill_projections(Pl, Pr);
    if (proj_l) memcpy(proj_l, Pl, sizeof Pl);
    if (proj_r) memcpy(proj_r, Pr, sizeof Pr);

    double* rot = (double*)malloc(sizeof(double) * 9 * nc);
    double* pos = (double*)malloc(sizeof(double) * 3 * nc);
    uint32_t* fill = (uint32_t*)calloc((size_t)nc + 1, sizeof(uint32_t));
    if (!rot || !pos || !fill) { free(rot); free(pos); free(fill); return -2; }
    truth_path(p, rot, pos);

    for (uint32_t k = 0; k < nc; k++) {
        const double* R = rot + 9 * (size_t)k;
        const double* t = pos + 3 * (size_t)k;
        if (true_poses_wc) {
            double* o = true_poses_wc + 16 * (size_t)k;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) o[i * 4 + j] = R[i * 3 + j];
                o[i * 4 + 3] = t[i];
            }
            o[12] = o[13] = o[14] = 0.0; o[15] = 1.0;
        }
        if (poses_wc) {
            double w[3], dt[3], E[9], Ri[9], ti[3];
            for (int a = 0; a < 3; a++) {
                w[a] = p->pose_rot_sigma * soslam_synth_normal(p->seed, ST_POSE, (uint64_t)k * 6 + a);
                dt[a] = p->pose_trans_sigma * soslam_synth_normal(p->seed, ST_POSE, (uint64_t)k * 6 + 3 + a);
            }
            so3_exp(w, E);
            mat3_mul(R, E, Ri);
            for (int i = 0; i < 3; i++) ti[i] = R[i * 3 + 0] * dt[0] + R[i * 3 + 1] * dt[1] + R[i * 3 + 2] * dt[2] + t[i];
            float* o = poses_wc + 16 * (size_t)k;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) o[i * 4 + j] = (float)Ri[i * 3 + j];
                o[i * 4 + 3] = (float)ti[i];
            }
            o[12] = o[13] = o[14] = 0.0f; o[15] = 1.0f;
        }
    }

    /* pass 1: per-frame observation counts -> frame-major offsets */
    const int want_obs = (obs_frame || obs_point || obs_uv);
    if (want_obs) {
        for (uint32_t j = 0; j < np; j++) {
            uint32_t c0 = first_cam(p, j), len = track_length(p, j, c0);
            for (uint32_t k = 0; k < len; k++) fill[c0 + k + 1]++;
        }
        for (uint32_t k = 0; k < nc; k++) fill[k + 1] += fill[k];
    }

    /* pass 2: points and observations (points ascending => per-frame point order ascending) */
    for (uint32_t j = 0; j < np; j++) {
        uint32_t c0 = first_cam(p, j), len = track_length(p, j, c0);
        double zlo = p->spacing * (double)(len - 1) + 4.0;
        if (zlo < 6.0) zlo = 6.0;
        double loc[3];
        loc[0] = -15.0 + 30.0 * soslam_synth_uniform(p->seed, ST_PX, j);
        loc[1] = -2.0 + 5.0 * soslam_synth_uniform(p->seed, ST_PY, j);
        loc[2] = zlo + (60.0 - zlo) * soslam_synth_uniform(p->seed, ST_PZ, j);
        const double* R0 = rot + 9 * (size_t)c0;
        const double* t0 = pos + 3 * (size_t)c0;
        double X[3];
        for (int i = 0; i < 3; i++) X[i] = R0[i * 3 + 0] * loc[0] + R0[i * 3 + 1] * loc[1] + R0[i * 3 + 2] * loc[2] + t0[i];
        if (true_points) for (int i = 0; i < 3; i++) true_points[3 * (size_t)j + i] = X[i];
        if (points) {
            double sc = 1.0 + p->depth_noise * soslam_synth_normal(p->seed, ST_DEPTH, j);
            for (int i = 0; i < 3; i++)
                points[3 * (size_t)j + i] =
                    (float)(R0[i * 3 + 0] * loc[0] * sc + R0[i * 3 + 1] * loc[1] * sc + R0[i * 3 + 2] * loc[2] * sc + t0[i]);
        }
        if (!want_obs) continue;
        for (uint32_t k = 0; k < len; k++) {
            uint32_t c = c0 + k;
            const double* R = rot + 9 * (size_t)c;
            const double* t = pos + 3 * (size_t)c;
            double d[3] = {X[0] - t[0], X[1] - t[1], X[2] - t[2]};
            double pc[3];
            for (int i = 0; i < 3; i++) pc[i] = R[0 * 3 + i] * d[0] + R[1 * 3 + i] * d[1] + R[2 * 3 + i] * d[2];
            double uv[4];
            project(Pl, pc, &uv[0], &uv[1]);
            project(Pr, pc, &uv[2], &uv[3]);
            uint64_t idx = (uint64_t)j * MAX_TRACK + k;
            int outlier = soslam_synth_uniform(p->seed, ST_OSEL, idx) < p->outlier_frac;
            for (int a = 0; a < 4; a++) {
                uv[a] += p->pixel_sigma * soslam_synth_normal(p->seed, ST_NOISE, idx * 4 + a);
                if (outlier) uv[a] += p->outlier_px * (2.0 * soslam_synth_uniform(p->seed, ST_OVAL, idx * 4 + a) - 1.0);
            }
            uint32_t slot = fill[c]++;
            if (obs_frame) obs_frame[slot] = c;
            if (obs_point) obs_point[slot] = j;
            if (obs_uv) for (int a = 0; a < 4; a++) obs_uv[4 * (size_t)slot + a] = (float)uv[a];
        }
    }
    free(rot); free(pos); free(fill);
    return 0;
}

/* ---- pose-graph workload ------------------------------------------------- */

int soslam_synth_pg_config(int config, soslam_synth_pg_params* p)
{
    if (!p) return -1;
    memset(p, 0, sizeof *p);
    p->seed = SOSLAM_SYNTH_SEED;
    p->min_gap = 50;
    p->step = 1.0;
    p->radius = 3.0;
    p->meas_trans_sigma = 0.02;
    p->meas_rot_sigma = 0.005;
    p->init_trans_sigma = 0.01;
    p->init_rot_sigma = 0.001;
    switch (config) {
    case 5: p->n_node = 5000; p->n_loop_max = 15001; p->row_len = 100; return 0;
    case 6: p->n_node = 200; p->n_loop_max = 300; p->row_len = 60; p->min_gap = 50; return 0; /* small test graph */
    default: return -1;
    }
}

static void pg_truth(const soslam_synth_pg_params* p, uint32_t i, double* R, double* t)
{
    uint32_t row = i / p->row_len, col = i % p->row_len;
    int fwd = (row % 2u) == 0u;
    double x = p->step * (double)(fwd ? col : (p->row_len - 1 - col));
    double y = p->step * (double)row;
    t[0] = x; t[1] = y;
    t[2] = 0.5 * sin(0.08 * x) + 0.3 * cos(0.11 * y);
    double yaw = fwd ? 0.0 : 3.14159265358979323846;
    double rz[9], ry[9], rx[9], m[9];
    rot_z(yaw + 0.05 * sin(0.03 * (double)i), rz);
    rot_y(-0.04 * cos(0.08 * x), ry);
    rot_x(0.03 * sin(0.05 * (double)i), rx);
    mat3_mul(rz, ry, m);
    mat3_mul(m, rx, R);
}

/* Z = Ti^-1 Tj, then right-perturbed by noise drawn from stream st at index base */
static void pg_relative(const double* Ri, const double* ti, const double* Rj, const double* tj,
                        uint64_t seed, uint64_t st, uint64_t base, double rs, double ts,
                        double* Rz, double* tz)
{
    double Rit[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rit[a * 3 + b] = Ri[b * 3 + a];
    double Rr[9], d[3] = {tj[0] - ti[0], tj[1] - ti[1], tj[2] - ti[2]}, tr[3];
    mat3_mul(Rit, Rj, Rr);
    for (int a = 0; a < 3; a++) tr[a] = Rit[a * 3 + 0] * d[0] + Rit[a * 3 + 1] * d[1] + Rit[a * 3 + 2] * d[2];
    double w[3], dt[3], E[9];
    for (int a = 0; a < 3; a++) {
        w[a] = rs * soslam_synth_normal(seed, st, base * 6 + a);
        dt[a] = ts * soslam_synth_normal(seed, st, base * 6 + 3 + a);
    }
    so3_exp(w, E);
    mat3_mul(Rr, E, Rz);
    for (int a = 0; a < 3; a++) tz[a] = Rr[a * 3 + 0] * dt[0] + Rr[a * 3 + 1] * dt[1] + Rr[a * 3 + 2] * dt[2] + tr[a];
}

static void pack_tq(const double* R, const double* t, double* out)
{
    double q[4];
    mat3_to_quat(R, q);
    double qf[4], n = 0.0;
    for (int a = 0; a < 4; a++) { qf[a] = f32r(q[a]); n += qf[a] * qf[a]; }
    (void)n;
    out[0] = f32r(t[0]); out[1] = f32r(t[1]); out[2] = f32r(t[2]);
    out[3] = qf[0]; out[4] = qf[1]; out[5] = qf[2]; out[6] = qf[3];
}

static uint64_t pg_loop_candidates(const soslam_synth_pg_params* p, uint64_t keep_total, uint64_t cand_total,
                                   uint32_t* e_from, uint32_t* e_to, uint32_t edge0)
{
    /* enumerate (i later, j earlier) with i-j >= min_gap and |ti-tj| <= radius, i ascending, j ascending;
       when keep_total>0 keep candidate t iff floor((t+1)K/N) > floor(tK/N) */
    uint64_t t = 0, kept = 0;
    const uint32_t rl = p->row_len;
    for (uint32_t i = 0; i < p->n_node; i++) {
        double Ri[9], ti[3];
        pg_truth(p, i, Ri, ti);
        uint32_t row = i / rl;
        uint32_t r0 = row >= 4 ? row - 4 : 0;
        for (uint32_t j = r0 * rl; j + p->min_gap <= i; j++) {
            double Rj[9], tj[3];
            pg_truth(p, j, Rj, tj);
            double dx = ti[0] - tj[0], dy = ti[1] - tj[1], dz = ti[2] - tj[2];
            if (dx * dx + dy * dy + dz * dz > p->radius * p->radius) continue;
            if (keep_total) {
                uint64_t a = ((t + 1) * keep_total) / cand_total, b = (t * keep_total) / cand_total;
                if (a > b) {
                    if (e_from) e_from[edge0 + kept] = i;
                    if (e_to) e_to[edge0 + kept] = j;
                    kept++;
                }
            }
            t++;
        }
    }
    return keep_total ? kept : t;
}

int soslam_synth_pg_count(const soslam_synth_pg_params* p, uint32_t* n_edge)
{
    if (!p || !n_edge || p->n_node < 2 || p->row_len == 0) return -1;
    uint64_t cand = pg_loop_candidates(p, 0, 0, NULL, NULL, 0);
    uint64_t loops = cand < p->n_loop_max ? cand : p->n_loop_max;
    *n_edge = (uint32_t)(p->n_node - 1 + loops);
    return 0;
}

int soslam_synth_pg_generate(const soslam_synth_pg_params* p, double* est,
                             uint32_t* e_from, uint32_t* e_to, double* meas, double* true_est)
{
    if (!p || p->n_node < 2 || p->row_len == 0) return -1;
    const uint32_t n = p->n_node;
    uint64_t cand = pg_loop_candidates(p, 0, 0, NULL, NULL, 0);
    uint64_t loops = cand < p->n_loop_max ? cand : p->n_loop_max;
    uint32_t* ef = e_from, *et = e_to;
    uint32_t* tmp_f = NULL, *tmp_t = NULL;
    if (!ef) { tmp_f = (uint32_t*)malloc(sizeof(uint32_t) * (n - 1 + loops)); ef = tmp_f; }
    if (!et) { tmp_t = (uint32_t*)malloc(sizeof(uint32_t) * (n - 1 + loops)); et = tmp_t; }
    if (!ef || !et) { free(tmp_f); free(tmp_t); return -2; }
    for (uint32_t i = 1; i < n; i++) { ef[i - 1] = i - 1; et[i - 1] = i; }
    if (loops) pg_loop_candidates(p, loops, cand, ef, et, n - 1);

    if (true_est) {
        for (uint32_t i = 0; i < n; i++) {
            double R[9], t[3];
            pg_truth(p, i, R, t);
            pack_tq(R, t, true_est + 7 * (size_t)i);
        }
    }
    if (est) {
        double Rc[9], tc[3];
        pg_truth(p, 0, Rc, tc);
        pack_tq(Rc, tc, est);
        for (uint32_t i = 1; i < n; i++) {
            double Ra[9], ta[3], Rb[9], tb[3], Rz[9], tz[3], Rn[9], tn[3];
            pg_truth(p, i - 1, Ra, ta);
            pg_truth(p, i, Rb, tb);
            pg_relative(Ra, ta, Rb, tb, p->seed, ST_PG_INIT, i, p->init_rot_sigma, p->init_trans_sigma, Rz, tz);
            mat3_mul(Rc, Rz, Rn);
            for (int a = 0; a < 3; a++) tn[a] = Rc[a * 3 + 0] * tz[0] + Rc[a * 3 + 1] * tz[1] + Rc[a * 3 + 2] * tz[2] + tc[a];
            memcpy(Rc, Rn, sizeof Rn); memcpy(tc, tn, sizeof tn);
            pack_tq(Rc, tc, est + 7 * (size_t)i);
        }
    }
    if (meas) {
        uint32_t ne = (uint32_t)(n - 1 + loops);
        for (uint32_t e = 0; e < ne; e++) {
            double Ra[9], ta[3], Rb[9], tb[3], Rz[9], tz[3];
            pg_truth(p, ef[e], Ra, ta);
            pg_truth(p, et[e], Rb, tb);
            pg_relative(Ra, ta, Rb, tb, p->seed, ST_PG_MEAS, e, p->meas_rot_sigma, p->meas_trans_sigma, Rz, tz);
            pack_tq(Rz, tz, meas + 7 * (size_t)e);
        }
    }
    free(tmp_f); free(tmp_t);
    return 0;
}
