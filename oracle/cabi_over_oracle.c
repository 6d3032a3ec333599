/*
 * cabi_over_oracle.c - TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * The subset of include/soslam_ba.h that the host shim's BundleAdjuster (stereo_orb_slam_amd/host/bundle_adjuster.cpp)
 * calls, implemented over the CPU oracle.  Linked with the UNCHANGED shim and demo sources it gives `ba_demo_oracle`: the
 * reference's call schedule (/root/reference/src/slam.cpp:121-129) with every container side effect of the shim
 * (/root/reference/src/camera_frame.h:32-72, float32 write-back) but the oracle as the solver - the thing the GPU-backed
 * ba_demo is compared with in tests/test_host_shim_gpu.py::test_slam_schedule_matches_the_oracle.  Nothing under
 * stereo_orb_slam_amd/ links or loads this file.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/soslam_ba.h"
#include "ba_oracle.h"

struct soslam_ba {
    soslam_ba_options opt;
    double pl[12], pr[12];
    uint32_t n_cam, n_pt, n_obs;
    uint32_t *obs_cam, *obs_pt;
    float* uv;
    uint8_t* fixed;
    double *cams, *pts;
};

const char* soslam_version(void) { return "oracle behind the soslam_ba C ABI (test infrastructure)"; }
const char* soslam_status_string(int status) { return status == SOSLAM_OK ? "ok" : "oracle failure"; }
const char* soslam_last_error(void) { return ""; }

void soslam_ba_options_default(soslam_ba_options* o)
{
    /* the values of include/soslam_ba.h's comments = /root/reference/src/params.h:34-47 + Ceres defaults */
    memset(o, 0, sizeof *o);
    o->max_iterations = 50; o->check_termination = 1; o->linear_solver = SOSLAM_SOLVER_AUTO; o->pcg_max_iterations = 500;
    o->pcg_tolerance = 1e-10; o->huber_delta = 1.0; o->lower_bound = -10000.0; o->upper_bound = 10000.0;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->parameter_tolerance = 1e-8; o->function_tolerance = 1e-16;
    o->gradient_tolerance = 1e-16; o->max_solver_time_seconds = 0.0; o->jacobi_scaling = 1; o->device = -1;
}

int soslam_ba_create(const soslam_ba_options* opts, soslam_ba** out)
{
    soslam_ba* h = (soslam_ba*)calloc(1, sizeof *h);
    if (!h) return SOSLAM_ERR_HIP;
    if (opts) h->opt = *opts; else soslam_ba_options_default(&h->opt);
    *out = h;
    return SOSLAM_OK;
}

int soslam_ba_set_options(soslam_ba* h, const soslam_ba_options* opts) { h->opt = *opts; return SOSLAM_OK; }

static void drop_problem(soslam_ba* h)
{
    free(h->obs_cam); free(h->obs_pt); free(h->uv); free(h->fixed); free(h->cams); free(h->pts);
    h->obs_cam = h->obs_pt = NULL; h->uv = NULL; h->fixed = NULL; h->cams = h->pts = NULL;
}

void soslam_ba_destroy(soslam_ba* h)
{
    if (!h) return;
    drop_problem(h);
    free(h);
}

int soslam_ba_set_projection(soslam_ba* h, const double* pl, const double* pr)
{
    memcpy(h->pl, pl, sizeof h->pl); memcpy(h->pr, pr, sizeof h->pr);
    return SOSLAM_OK;
}

static void* dup(const void* p, size_t n) { void* q = malloc(n ? n : 1); if (q && n) memcpy(q, p, n); return q; }

int soslam_ba_set_problem(soslam_ba* h, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs, const uint32_t* obs_cam,
                          const uint32_t* obs_pt, const float* obs_uv, const uint8_t* cam_fixed)
{
    drop_problem(h);
    h->n_cam = n_cam; h->n_pt = n_pt; h->n_obs = n_obs;
    h->obs_cam = (uint32_t*)dup(obs_cam, sizeof(uint32_t) * n_obs);
    h->obs_pt = (uint32_t*)dup(obs_pt, sizeof(uint32_t) * n_obs);
    h->uv = (float*)dup(obs_uv, sizeof(float) * 4 * n_obs);
    h->fixed = (uint8_t*)calloc(n_cam ? n_cam : 1, 1);
    if (cam_fixed) memcpy(h->fixed, cam_fixed, n_cam);
    h->cams = (double*)calloc((size_t)n_cam * 6 + 1, sizeof(double));
    h->pts = (double*)calloc((size_t)n_pt * 3 + 1, sizeof(double));
    return (h->obs_cam && h->obs_pt && h->uv && h->fixed && h->cams && h->pts) ? SOSLAM_OK : SOSLAM_ERR_HIP;
}

int soslam_ba_set_state(soslam_ba* h, const double* poses, const double* points)
{
    memcpy(h->cams, poses, sizeof(double) * 6 * h->n_cam);
    if (h->n_pt) memcpy(h->pts, points, sizeof(double) * 3 * h->n_pt);
    return SOSLAM_OK;
}

int soslam_ba_get_state(soslam_ba* h, double* poses, double* points)
{
    if (poses) memcpy(poses, h->cams, sizeof(double) * 6 * h->n_cam);
    if (points && h->n_pt) memcpy(points, h->pts, sizeof(double) * 3 * h->n_pt);
    return SOSLAM_OK;
}

/* the sharded entry points of the shim, for one rank (the oracle has no collective): enough to link the unchanged shim */
int soslam_rccl_get_unique_id(void* id128) { memset(id128, 0, 128); return SOSLAM_OK; }
int soslam_ba_init_rccl(soslam_ba* h, const void* id128, int32_t rank, int32_t world) { (void)h; (void)id128; return (rank == 0 && world == 1) ? SOSLAM_OK : SOSLAM_ERR_COMM; }
int soslam_ba_set_covisibility(soslam_ba* h, uint64_t n, const uint32_t* a, const uint32_t* b) { (void)h; (void)n; (void)a; (void)b; return SOSLAM_OK; }
int soslam_ba_agree_status(soslam_ba* h, int local_status, int* agreed) { (void)h; if (agreed) *agreed = local_status; return SOSLAM_OK; }   /* one rank */
void soslam_ba_shard_range(uint32_t n_pt, int32_t rank, int32_t world, uint32_t* begin, uint32_t* end)
{
    if (world < 1) world = 1;
    if (begin) *begin = (uint32_t)(((uint64_t)rank * n_pt) / (uint64_t)world);
    if (end) *end = (uint32_t)(((uint64_t)(rank + 1) * n_pt) / (uint64_t)world);
}
int soslam_ba_get_state_global(soslam_ba* h, double* poses, uint32_t n_pt_global, uint32_t shard_begin, double* points_global)
{
    if (shard_begin != 0 || n_pt_global != h->n_pt) return SOSLAM_ERR_INVALID_ARGUMENT;
    return soslam_ba_get_state(h, poses, points_global);
}

int soslam_ba_solve(soslam_ba* h, soslam_ba_summary* summary)
{
    oracle_ba_options o;
    oracle_ba_options_default(&o);
    o.max_iterations = h->opt.max_iterations; o.check_termination = h->opt.check_termination; o.huber_delta = h->opt.huber_delta;
    o.lower_bound = h->opt.lower_bound; o.upper_bound = h->opt.upper_bound; o.initial_radius = h->opt.initial_radius;
    o.max_radius = h->opt.max_radius; o.min_radius = h->opt.min_radius; o.min_relative_decrease = h->opt.min_relative_decrease;
    o.min_lm_diagonal = h->opt.min_lm_diagonal; o.max_lm_diagonal = h->opt.max_lm_diagonal;
    o.parameter_tolerance = h->opt.parameter_tolerance; o.function_tolerance = h->opt.function_tolerance;
    o.gradient_tolerance = h->opt.gradient_tolerance; o.jacobi_scaling = h->opt.jacobi_scaling; o.num_threads = 4;
    oracle_ba_summary s;
    memset(&s, 0, sizeof s);
    const int rc = oracle_ba_solve(h->n_cam, h->n_pt, h->n_obs, h->obs_cam, h->obs_pt, h->uv, h->cams, h->pts, h->pl, h->pr, h->fixed, &o,
                                   &s, NULL);
    if (summary) {
        memset(summary, 0, sizeof *summary);
        summary->initial_cost = s.initial_cost; summary->final_cost = s.final_cost; summary->iterations = s.iterations;
        summary->accepted = s.accepted; summary->termination = s.termination; summary->line_search_steps = s.line_search_steps;
        summary->solve_seconds = s.solve_seconds; summary->setup_seconds = s.setup_seconds;
    }
    return rc == 0 ? SOSLAM_OK : SOSLAM_ERR_LINEAR_SOLVER;
}
