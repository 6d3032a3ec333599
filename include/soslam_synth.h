/*
 * soslam_synth.h - deterministic synthetic workloads for the bundle-adjustment
 * and pose-graph hot path.
 *
 * The reference ships no data set, camera file or fixture (SURVEY.md section 0,
 * fact 5), so every problem this repository measures is produced by the one
 * counter-based generator declared here.  The generator is a workload source,
 * not part of the solver and not part of the oracle; the same C file is linked
 * into the product library and into oracle/_build so both sides see identical
 * bytes.  A numpy mirror (stereo_orb_slam_amd/synth.py) reproduces the raw
 * 64-bit stream exactly and the derived floats to 1 ulp of float32.
 *
 * Data conventions follow the reference containers:
 *   - poses are camera->world 4x4 float32 row-major, as Frame::GlobalPose()
 *     (/root/reference/src/camera_frame.h:29) and VisualOdometer::Dump
 *     (/root/reference/src/visual_odometer.cpp:453-461) hold them;
 *   - points are float32 xyz (/root/reference/src/map_point.h:44);
 *   - observations are {frame_id, point_id, u_l, v_l, u_r, v_r} float32, frame
 *     by frame, as Dump writes constraints.txt
 *     (/root/reference/src/visual_odometer.cpp:493-502);
 *   - projections are row-major 3x4, widened from float32 as
 *     InitializeStereoReprojectionError does (/root/reference/src/slam.cpp:176-209).
 */
#ifndef SOSLAM_SYNTH_H
#define SOSLAM_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOSLAM_SYNTH_SEED 20241004ull

/* Track-length law of a synthetic BA problem. */
enum {
    SOSLAM_TRACK_FIXED = 0,     /* every point is seen by `track_len` consecutive cameras */
    SOSLAM_TRACK_GEOMETRIC = 1  /* geometric with mean `track_len`, capped by remaining frames */
};

typedef struct soslam_synth_ba_params {
    uint64_t seed;
    uint32_t n_cam;
    uint32_t n_pt;
    uint32_t track_mode;      /* SOSLAM_TRACK_* */
    uint32_t track_len;       /* fixed length, or mean of the geometric law */
    double   spacing;         /* metres between consecutive cameras */
    double   curvature;       /* rad per metre of yaw along the path */
    double   pixel_sigma;     /* Gaussian pixel noise per coordinate */
    double   outlier_frac;    /* fraction of observations that get +-outlier_px uniform error */
    double   outlier_px;
    double   pose_rot_sigma;  /* initial pose perturbation, rad */
    double   pose_trans_sigma;/* initial pose perturbation, m */
    double   depth_noise;     /* relative depth noise of initial points */
} soslam_synth_ba_params;

/* Raw generator: 64-bit value `index` of stream `stream` under `seed`. */
uint64_t soslam_synth_u64(uint64_t seed, uint64_t stream, uint64_t index);
/* Uniform in [0,1) and standard normal (Box-Muller on two uniforms) of the same counter. */
double soslam_synth_uniform(uint64_t seed, uint64_t stream, uint64_t index);
double soslam_synth_normal(uint64_t seed, uint64_t stream, uint64_t index);

/* BASELINE.json configs[0..2]: 1 = 10/2k/~8k, 2 = 100/20k/200k, 3 = 500/100k/1M. */
int soslam_synth_ba_config(int config, soslam_synth_ba_params* out);

/* Number of observations the parameters produce (exact). */
int soslam_synth_ba_count(const soslam_synth_ba_params* p, uint32_t* n_obs);

/*
 * Fill caller-allocated arrays.  Any output pointer may be NULL.
 *   poses_wc      n_cam*16 float32   initial camera->world, row-major
 *   points        n_pt*3   float32   initial positions
 *   obs_frame     n_obs    uint32    frame-major order (frame ascending, point ascending)
 *   obs_point     n_obs    uint32
 *   obs_uv        n_obs*4  float32   u_l v_l u_r v_r
 *   proj_l/proj_r 12       double    row-major 3x4, float32-representable
 *   true_poses_wc n_cam*16 double    ground truth (for diagnostics only)
 *   true_points   n_pt*3   double
 */
int soslam_synth_ba_generate(const soslam_synth_ba_params* p,
                             float* poses_wc, float* points,
                             uint32_t* obs_frame, uint32_t* obs_point, float* obs_uv,
                             double* proj_l, double* proj_r,
                             double* true_poses_wc, double* true_points);

/*
 * Pose graph of BASELINE.json configs[4]: n_node SE(3) nodes on a 3-D lawn-mower
 * path, n_node-1 odometry edges (i-1 -> i) followed by loop edges between nodes
 * at least `min_gap` apart in index and within `radius` metres, capped at
 * `n_loop_max`.  Estimates and measurements are [tx ty tz qx qy qz qw] float32-
 * representable doubles (the reference narrows both through Matrix4f,
 * /root/reference/src/pose_graph_optimizer.cpp:103-116,157-161).
 */
typedef struct soslam_synth_pg_params {
    uint64_t seed;
    uint32_t n_node;
    uint32_t n_loop_max;
    uint32_t min_gap;
    uint32_t row_len;      /* nodes per lawn-mower row */
    double   step;         /* metres between consecutive nodes */
    double   radius;       /* loop-edge search radius, metres */
    double   meas_trans_sigma;
    double   meas_rot_sigma;
    double   init_trans_sigma; /* drift per step accumulated into the initial estimates */
    double   init_rot_sigma;
} soslam_synth_pg_params;

int soslam_synth_pg_config(int config, soslam_synth_pg_params* out);
int soslam_synth_pg_count(const soslam_synth_pg_params* p, uint32_t* n_edge);
int soslam_synth_pg_generate(const soslam_synth_pg_params* p,
                             double* est /* n_node*7 */,
                             uint32_t* e_from, uint32_t* e_to, double* meas /* n_edge*7 */,
                             double* true_est /* n_node*7, may be NULL */);

#ifdef __cplusplus
}
#endif
#endif
