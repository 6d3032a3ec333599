// ba_kernels.h - launch wrappers of the BA device kernels (definitions in ba_kernels.hip).
// All pointers are device pointers; every launch is asynchronous on `s`.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_device.h"

namespace soslam {

constexpr int kTileObs = 1024;       // observations per linearize/cost workgroup (one camera per tile)
constexpr int kTileThreads = 256;    // lanes per tile: up to 4 observations each
constexpr int kTileVals = 28;        // 21 (J_c^T J_c upper) + 6 (J_c^T r) + 1 (rho)
constexpr int kArRow = 10;            // f64 per observation in the compact row: [G (xx xy xz yy yz zz)] 48 B + [h (3) | 0] 32 B
constexpr int kArG = 6, kArH = 4;     // ... held as TWO arrays: every pass after the linearisation gathers rows through the point-major
                                      // index, and most of them (back-substitution, the Schur products) want G only - with one
                                      // 80-B row per observation they fetched 1.7x the bytes they used (PMC, round 2)
struct CompactRows {
    double* g;                        // [n_obs][6]
    double* h;                        // [n_obs][4]
};
constexpr int kBatchObs = 128;       // observations staged per Schur batch (also the most one windowed point may have)
// points per Schur batch: the batch's point columns (3 each) are the k dimension of the window GEMM, and two
// [3 PB][6 KMAX + 1] f64 images must fit LDS next to the staged rows
constexpr int schur_batch_points(int kmax) { return 12; }
constexpr int kPointBlock = 256;     // threads per block of the per-point kernels
constexpr int kBacksubLanes = 4;     // lanes that share one point in ba_backsub
constexpr uint32_t backsub_blocks(uint32_t n_pt) { return (uint32_t)(((uint64_t)n_pt * kBacksubLanes + kPointBlock - 1) / kPointBlock); }

// scalar slots (device f64)
enum {
    SC_COST_X = 0,                                    // this rank's cost at the linearisation point
    SC_CAND_COST = 1, SC_MCC_PTS = 2, SC_STEP2_PTS = 3, SC_X2_PTS = 4, SC_GDOT_PTS = 5,  // summed over ranks
    SC_GMAX_PTS = 6,                                  // max over ranks; slots 2..6 are written as one group by sum5
    SC_STOP = 7,                                      // max over ranks (with SC_GMAX_PTS): a rank's vote to end the solve (time limit)
    SC_MCC_CAM = 8, SC_STEP2_CAM = 9, SC_X2_CAM = 10, SC_GDOT_CAM = 11, SC_GMAX_CAM = 12,  // replicated
    SC_LIN_ITERS = 13, SC_LIN_RESID = 14, SC_LIN_STATUS = 15,
    SC_SCHUR_STATUS = 16,                             // rank-local; spread through the candidate cost (launch_status_poison);
                                                      // next to the solver's slots: one memset clears all four per iteration
    SC_GATE = 17,                                     // 1: the device accepted the step and has linearised at the candidate
    SC_LS_COST = 18, SC_LS_DIR = 19, SC_LS_STEP2 = 20,   // line search trial: cost, direction . gradient, |x+ - x|^2 over the points
    SC_LS_DMAX = 21,                                  // max |delta_i| (LineSearchFunction::DirectionInfinityNorm)
    SC_COUNT = 22
};

struct Tile {      // one workgroup of ba_linearize / ba_cost
    uint32_t cam;
    uint32_t start;  // first observation (camera-major index)
    uint32_t count;  // <= kTileObs
    uint32_t pad;
};

struct SchurChunk {  // one workgroup of ba_schur
    uint32_t batch_begin, batch_end;
    uint32_t n_local;    // local cameras used (<= KMAX)
    uint32_t pad;
};

struct SchurBatch {
    uint32_t q_begin, q_end;  // point-major observation positions
    uint32_t p_begin, p_end;  // internal point range
    uint32_t full;            // 1: a whole batch of points, each observed by every camera of the window, none of them fixed -
                              // the batch overwrites every element of the LDS image it reads (no zeroing needed)
};
// A point seen by more free cameras than the widest Schur window (32), or with more observations than a batch
// holds, is eliminated by the long-track kernels: its W / Y blocks go to a scratch array and one 36-lane group per
// camera pair writes that pair's product to the slab.
struct LongPoint {
    uint32_t p;                 // internal point
    uint32_t lo_begin, lo_end;  // its free-camera observations in the long-observation arrays
    uint32_t pad;
};

struct LmDiag {     // what a kernel needs to rebuild the point damping
    double radius, lo, hi;
};

// Ceres' damping of one diagonal entry in the Jacobi-scaled problem, expressed in unscaled units
__device__ __forceinline__ double point_lambda(double cdiag, double s, const LmDiag& lm)
{
    const double s2 = s * s;
    return fmin(fmax(s2 * cdiag, lm.lo), lm.hi) / (lm.radius * s2);
}

// Camera damping as ba_cam_damp_kernel applies it, for a caller that folds it into a kernel of its own (the cyclic
// reduction's gather): lam = point_lambda(diag B, scale), lc = lam, diagonal of S += lam; scale set on the first pass.
struct CamDamp {
    const double* diagB;
    double* sc;
    double* lc;
    const int32_t* diag_block;
    LmDiag lm;
    int init_scale, jacobi;
    uint32_t n_free;
};

// coarse basis of the two-level PCG (free camera f: row f of P, 36 f64), aggregates given by row_agg / agg_ref (a camera index)
void launch_ba_coarse_basis(hipStream_t s, uint32_t n_free, const uint32_t* free_cam, const uint32_t* row_agg, const uint32_t* agg_ref,
                            const double* campre, double* P);

// campre[n_cam][kPoseStride]: per-camera rotation block consumed by launch_linearize / launch_cost
void launch_pose_prepare(hipStream_t s, uint32_t n_cam, const double* cams, double* campre);

void launch_linearize(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt,
                      const double* campre, const double* pts, const int32_t* cam_free, const Proj& P, double delta,
                      CompactRows ar, double* tile_part, const double* gate = nullptr /* != NULL: run only if *gate != 0 */);

// parity tests: residual (n_obs*4), J_c (n_obs*24), J_p (n_obs*12) of every observation, internal observation order
void launch_debug_rows(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt,
                       const double* campre, const double* pts, const int32_t* cam_free, const Proj& P, double delta,
                       double* r_out, double* jc_out, double* jp_out);

void launch_cost(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt,
                 const double* campre, const double* pts, const Proj& P, double delta, double* cost_part);

// out[0] = scale * sum_{i<n} in[i*stride + offset], one workgroup, fixed order (bitwise reproducible)
void launch_sum_strided(hipStream_t s, const double* in, uint32_t n, uint32_t stride, uint32_t offset, double scale, double* out);
// per-workgroup partials [n][5] = {sum, sum, sum, sum, max} -> out[0..4]
// n <= 64 doubles to pinned host memory, then *host_seq = seq (release at system scope); src[clear_first .. +clear_n) are
// zeroed afterwards (the status words of the next iteration)
void launch_publish(hipStream_t s, double* src, int n, int clear_first, int clear_n, double* host_dst, unsigned long long* host_seq,
                    unsigned long long seq);
// The iteration's final sums in one launch: out5[0..4] = the five step scalars of the back-substitution partials (sums,
// last one a max), out_cam5 the same of the camera-update partials, out_cost = 0.5 sum cost_part; host_dst != NULL: then the publication (see launch_publish)
// gate != NULL: *gate = the LM acceptance test (status words all zero, finite positive model change, (x_cost - candidate
// cost) / model change > min_relative_decrease) if gate_enabled, else 0 - for launches enqueued behind this one
// stop_vote: this rank's vote to end the solve, written to out5[5] (SC_STOP follows the five step scalars)
void launch_step_sums(hipStream_t s, const double* part5, uint32_t n5, double* out5, const double* cam5, uint32_t n_cam5, double* out_cam5,
                      const double* cost_part, uint32_t n_cost, double* out_cost, double* gate, const double* status, double x_cost,
                      double min_relative_decrease, int gate_enabled, double stop_vote, double* pub_src, int n_pub, int clear_first,
                      int clear_n, double* host_dst, unsigned long long* host_seq, unsigned long long seq, int armijo_in_gate = 1);
// Multi-rank jobs: the acceptance test and the publication AFTER the step scalars were summed over the ranks - the same
// test on the same operands as launch_step_sums applies on one rank, read from the scalar slots (scal = the SC_* array)
void launch_gate_publish(hipStream_t s, double* scal, double x_cost, double min_relative_decrease, int gate_enabled, double* pub_src,
                         int n_pub, int clear_first, int clear_n, double* host_dst, unsigned long long* host_seq, unsigned long long seq,
                         int armijo_in_gate = 1);
void launch_sum5(hipStream_t s, const double* in, uint32_t n, double* out);

void launch_point_reduce(hipStream_t s, uint32_t n_pt, const uint32_t* pt_start, const uint32_t* pt_obs, const uint32_t* q_cam,
                         CompactRows ar, const double* campre, double* C, double* gp, const double* gate = nullptr);

void launch_point_scale(hipStream_t s, uint32_t n_pt, const double* C, int jacobi, double* sp);

// chunk windows into per-chunk slabs (chunk_slab[c] = offset of chunk c, layout [pair][36] then [camera][6])
constexpr int kS10PairsPerBatch = 120;   // (point, window slot) pairs of a batch of the ten-camera kernel: 12 points x 10 slots
void launch_schur(hipStream_t s, int kmax, uint32_t n_chunks, const SchurChunk* chunks, const SchurBatch* batches,
                  const uint32_t* chunk_slab, const uint32_t* chunk_cam /* [n_chunks][kmax] camera of each window slot */,
                  const uint32_t* pair_row /* kmax <= 10: [n_batches][120] row of (point pl, slot) at pl * n_local + slot, or ~0 */,
                  const uint32_t* pt_obs, const uint32_t* q_pt, const uint8_t* q_slot, CompactRows ar, const double* campre,
                  const double* pts, double* C, double* gp, const double* sp, LmDiag lm, double* Cinv,
                  double* ptfac /* [n_pt][12] scratch: L^-T and L^-1 g of the damped point blocks, position */, double* slab, double* scal,
                  const uint32_t* pt_start, const uint32_t* q_cam,
                  int point_blocks_from_rows /* kmax <= 10 only: the kernel forms C and gp itself (and writes them): no launch_point_reduce needed */);

// Multi-rank jobs: a rank whose point elimination failed (SC_SCHUR_STATUS) turns its share of the candidate cost
// into +inf before the scalars are summed, so every rank sees a non-finite candidate and rejects the step alike.
void launch_status_poison(hipStream_t s, double* scal);

// Small problems (at most 32 cameras - the reference's windows): ba_cam_update + ba_backsub + ba_cost in ONE launch.  Every workgroup
// forms the candidate cameras and their pose table itself; cost_part gets backsub_blocks(n_pt) entries (one per workgroup).
// sums != NULL: the workgroup that finishes last also sums the step scalars, tests acceptance and publishes (what launch_step_sums
// does in a launch of its own; same arguments); arrivals: one zero-initialised word the kernel leaves zero
struct StepSumsLaunch {
    double* out5; double* out_cam5; double* out_cost; double* gate; const double* status;
    double x_cost, min_relative_decrease; int gate_enabled; double stop_vote;
    double* pub_src; int n_pub, clear_first, clear_n; double* host_dst; unsigned long long* host_seq; unsigned long long seq; int armijo_in_gate;
};
bool apply_small_fits(uint32_t n_cam, uint32_t n_pt);
void launch_apply_small(hipStream_t s, uint32_t n_cam, const int32_t* cam_free, const double* cams, const double* dc_free, const double* lc,
                        const double* gc_red, const double* lin_resid, double* cams_out, double* dc_full, double* dcw, double* cam_part,
                        double* campre_c, uint32_t n_pt, const uint32_t* pt_start, const uint32_t* pt_obs, const uint32_t* q_cam, const double* ar,
                        const double* campre, const double* Cinv, const double* C, const double* gp, const double* sp, const double* pts, LmDiag lm,
                        double bound_lo, double bound_hi, double* pts_out, double* dp, double* part, const float4* uv, const Proj& P, double delta,
                        double* cost_part, unsigned int* arrivals, const StepSumsLaunch* sums);

// ---- small problems' uploads in one command ------------------------------------------------------------------------------
// The index of a window of a few thousand observations is ~35 arrays of a few KB: as separate copy commands they cost more than
// the index takes to build.  The host packs them into one pinned buffer (table of segments, then the 16-byte aligned payloads)
// and ONE kernel reads that buffer over the link and writes every array where it belongs (src_offset = ~0: fill with zeros).
// With payload == nullptr the offsets are source ADDRESSES: the same kernel gathers device arrays into pinned host memory.
struct PackedSeg { void* dst; uint64_t src_offset; uint64_t bytes; };
void launch_packed_scatter(hipStream_t s, const PackedSeg* table, int n_seg, const unsigned char* payload);

// ---- structure-only problems (every camera constant: BundleAdjuster::Optimize(n-1, n), /root/reference/src/slam.cpp:123) ----
// The problem decouples into independent 3-variable blocks under ONE trust region.  For problems of at most
// kPointsOnlyMax points a single workgroup does a whole LM iteration in one launch: linearise every point at x (J_p^T J_p,
// J_p^T r, cost), damp, solve, project the candidate onto the box, evaluate it, sum the step scalars, test acceptance,
// publish - what the general path spreads over a dozen launches (no Schur complement, no camera system).
constexpr uint32_t kPointsOnlyMax = 16384;
struct PointsStepArgs {
    uint32_t n_pt;
    const uint32_t* pt_start; const uint32_t* pt_obs; const uint32_t* q_cam;
    const float4* uv;
    const double* campre;            // pose table of the (constant) cameras
    const double* pts;               // x
    double* pts_out;                 // candidate
    double* dp;                      // full step (the line search's direction)
    double* C; double* gp; double* sp; double* Cinv;
    LmDiag lm;
    double huber_delta, bound_lo, bound_hi;
    int init_scale, jacobi;
    double* scal;                    // the SC_* array
    double* cost_x_out;              // where the host looks for the cost at x (the reduce payload's tail)
    double x_cost, min_relative_decrease;
    int gate_enabled;
};
void launch_points_step(hipStream_t s, const PointsStepArgs& a, const Proj& P, double* pub_src, int n_pub, int clear_first, int clear_n,
                        double* host_dst, unsigned long long* host_seq, unsigned long long seq);
// one line-search trial of a structure-only problem (candidate at step size `step`, its cost, direction . gradient, step norm
// and |dp|_inf into scal[SC_LS_*], published) in one launch
void launch_points_ls(hipStream_t s, const PointsStepArgs& a, const Proj& P, double step, double* pub_src, int n_pub, double* host_dst,
                      unsigned long long* host_seq, unsigned long long seq);

// The whole structure-only solve in one launch (ba_points.hip): trust-region controller, termination tests and the bounded
// problem's line search on the device; ceil(n_pt / 64) single-wave workgroups (kPointsOnlyMax / 64 at most: all resident).  In: the controller state of the handle; out (pinned host memory): one record of
// PSV_COUNT doubles, the iteration log (kPointsLogDoubles doubles per entry, laid out as soslam_ba_iteration), then `seq`.
enum { PSV_RADIUS = 0, PSV_DECREASE, PSV_X_COST, PSV_CUR, PSV_INVALID_RUN, PSV_ITERATIONS, PSV_ACCEPTED, PSV_TERMINATION, PSV_LS_STEPS,
       PSV_INITIAL_COST, PSV_N_LOG, PSV_ERROR, PSV_HAVE_INITIAL, PSV_PASSES, PSV_COUNT };
constexpr int kPointsLogDoubles = 9;
// termination codes as include/soslam_ba.h numbers them (checked where both are visible: ba_solver.hip)
enum { kPointsTermMaxIterations = 0, kPointsTermParameter = 1, kPointsTermFunction = 2, kPointsTermGradient = 3, kPointsTermMinRadius = 4,
       kPointsTermInvalid = 5, kPointsTermTime = 6 };
struct PointsSolveCtl {
    double* pts[2];                  // x is pts[cur], the candidate pts[cur ^ 1]
    int cur, invalid_run, init_scale, x_cost_known;
    double radius, decrease_factor, x_cost;
    int max_it, check, constrained;
    double lm_lo, lm_hi;
    double min_radius, max_radius, min_relative_decrease, gradient_tolerance, parameter_tolerance, function_tolerance;
    long long max_ticks;             // wall-clock limit in ticks of the 100 MHz constant clock; 0: none
    double* log;                     // device scratch, (max_it + 1) entries
    // the grid barrier / reduction of the single-wave workgroups (one point per lane): records [2][n_wg][16], a counter that only
    // ever grows (sync_base = its value when this launch starts)
    int n_wg;
    double* sync_records; unsigned long long* sync_counter; unsigned long long sync_base;
    double* host_record; double* host_log;
    unsigned long long* host_seq; unsigned long long seq;
};
void launch_points_solve(hipStream_t s, const PointsStepArgs& a, const Proj& P, const PointsSolveCtl& c);

// ---- Ceres' line search on bounded problems (TrustRegionMinimizer::DoLineSearch; see run_lm) ----------------------------
// trial point x+ = Plus(x, a delta): cameras x + a dc, points projected onto the box; ls_part[block][2] = {|x+ - x|^2 of the
// block's points, max |delta_i| of the block}
void launch_ls_candidate(hipStream_t s, uint32_t n_cam, uint32_t n_pt, const double* cams, const double* pts, const double* dc_full,
                         const double* dp, double a, double bound_lo, double bound_hi, double* cams_out, double* pts_out, double* ls_part,
                         double* campre_c /* the trial's pose table, written by the same launch */);
inline uint32_t ls_candidate_blocks(uint32_t n_pt) { return (n_pt * 3 + 255) / 256 + 1; }
// cost and direction . gradient at the trial point (campre_c, pts_c = its pose table and points; the direction is the FULL
// step dc_full / dp): per tile [rho sum, sum of r~^T (J~_c dc + J~_p dp)] into tile_part2[tile][2]
void launch_ls_eval(hipStream_t s, uint32_t n_tiles, const Tile* tiles, const float4* uv, const uint32_t* obs_pt, const double* campre_c,
                    const double* pts_c, const double* dc_full, const double* dp, const int32_t* cam_free, const Proj& P, double delta,
                    double* tile_part2);
// the sums of a trial into scal[SC_LS_COST .. SC_LS_DMAX]; host_dst != NULL: then the publication (see launch_publish)
void launch_ls_sums(hipStream_t s, const double* tile_part2, uint32_t n_tiles, const double* ls_part, uint32_t n_ls_part, double* scal,
                    double* pub_src, int n_pub, double* host_dst, unsigned long long* host_seq, unsigned long long seq);
// dst[i] = src[i], i < n; either side may be pinned host memory mapped into the device (host-collective staging)
void launch_copy_f64(hipStream_t s, double* dst, const double* src, uint64_t n);
// *host_seq = seq (pinned host memory) after everything enqueued before it has completed
void launch_flag(hipStream_t s, unsigned long long* host_seq, unsigned long long seq);

// long-track points (see LongPoint): damped point block inverse, W / Y of every free-camera observation into
// wy[lo][36] and Y g into the slab at lo_cam_off[lo]; then Y_a W_b^T of every listed pair into the slab at pair_off
void launch_schur_long(hipStream_t s, uint32_t n_long, const LongPoint* long_pts, const uint32_t* lo_row, const uint32_t* lo_cam,
                       const uint32_t* lo_cam_off, uint32_t n_pairs, const uint32_t* pair_a, const uint32_t* pair_b,
                       const uint32_t* pair_off, const double* ar, const double* campre, const double* pts, const double* C,
                       const double* gp, const double* sp, LmDiag lm, double* Cinv, double* wy, double* slab, double* scal);

// S = B - sum(slabs), rhs = -g_c + sum(slab rhs parts) through host-built contribution lists (fixed order);
// exports diag(B) and g_c next to them for the all-reduce
void launch_schur_reduce(hipStream_t s, uint32_t n_blocks, uint32_t n_free, const uint32_t* blk_ptr, const uint32_t* blk_off,
                         const uint32_t* cam_ptr, const uint32_t* cam_off, const uint32_t* blk_row, const uint32_t* blk_col,
                         const uint32_t* free_cam /* camera of each free index */, const double* campre, const double* slab,
                         const uint32_t* cam_tile_start, const double* tile_part /* ba_linearize's per-tile camera sums: the kernel
                         forms each camera's own block B and gradient g_c from them (T applied once per camera) */,
                         double* S, double* rhs, double* diagB, double* gc_red,
                         const double* cost_in, double* cost_out /* cost_out = cost_in: this rank's cost joins the reduce payload */,
                         const CamDamp* damp = nullptr /* one rank, dense solve: the camera damping too (no ba_cam_damp launch) */);

void launch_cam_damp(hipStream_t s, uint32_t n_free, const double* diagB, double* sc, int init_scale, int jacobi,
                     LmDiag lm, const int32_t* diag_block, double* S, double* lc);

// cam_part[cam_update_blocks(n_cam)][5]: per-workgroup partials of the camera share of the step scalars (launch_step_sums
// adds them into SC_MCC_CAM .. SC_GMAX_CAM)
constexpr int kCamUpdateCams = 42;   // whole cameras per workgroup of ba_cam_update (252 of its 256 lanes)
inline uint32_t cam_update_blocks(uint32_t n_cam) { return (n_cam + kCamUpdateCams - 1) / kCamUpdateCams; }
void launch_cam_update(hipStream_t s, uint32_t n_cam, const int32_t* cam_free, const double* cams,
                       const double* dc_free, const double* lc, const double* gc_red, const double* lin_resid, const double* campre,
                       double* cams_out, double* dc_full, double* dcw /* [n_cam][6]: M dc_rot | dc_t */, double* cam_part,
                       double* campre_c /* != NULL: the candidate's pose table, [n_cam][kPoseStride] */);

void launch_backsub(hipStream_t s, uint32_t n_pt, const uint32_t* pt_start, const uint32_t* pt_obs, const uint32_t* q_cam,
                    const double* ar, const double* campre, const double* dcw,
                    const double* Cinv, const double* C, const double* gp, const double* sp, const double* pts,
                    LmDiag lm, double bound_lo, double bound_hi, double* pts_out, double* dp, double* part);

}  // namespace soslam
