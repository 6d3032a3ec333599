// ba_solver.hip - host side of libsoslam_ba: index construction, device state, the Levenberg-Marquardt
// controller and the C ABI of include/soslam_ba.h.
//
// The controller restates Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy for the problem
// BundleAdjuster::Optimize builds (/root/reference/src/bundle_adjuster.cpp:39-118):
// all arithmetic on problem-sized data runs in the HIP kernels of ba_kernels.hip / linsolve.hip; the host
// reads back one small block of scalars per iteration and takes the accept/reject decision.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <complex>
#include <thread>
#include <cstdlib>
#include <iterator>
#include <memory>
#include <numeric>

#include "../host/mat4f.h"
#include "ba_kernels.h"
#include "common.h"
#include "linsolve.h"
#include "rccl_leg.h"

namespace soslam {

static thread_local std::string g_last_error;

void set_last_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

namespace {

double now_sec()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr int kDenseAutoLimit = 1200;   // camera dof up to which AUTO picks the dense Cholesky
constexpr uint32_t kPcgMultiMinRows = 64;   // below this one workgroup does a whole PCG iteration faster than three launches

}  // namespace

}  // namespace soslam

using namespace soslam;

struct soslam_ba {
    soslam_ba_options opt{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int device = 0;
    Proj proj{};
    bool have_proj = false, have_problem = false, have_state = false;

    // dimensions
    uint32_t n_cam = 0, n_pt = 0, n_obs = 0, n_free = 0, n_tiles = 0, n_chunks = 0, n_batches = 0, n_blocks = 0;
    uint32_t n_point_blocks = 0;
    int kmax = 16;
    int bw = 0;                         // block half-bandwidth of the reduced camera matrix
    bool pcg_band = false;              // PCG preconditioned by the band factor
    bool use_cr = false;                // band factor by block cyclic reduction (bw <= kCrBandMax)
    bool off_band = false;              // some blocks of S lie outside the factored band bw (loop closures): they stay in the matvec only
    int bw_full = 0;                    // the largest block offset when off_band
    double comp_share = 0.0;            // off_band: share of the left-out blocks' absolute row sums added to the factored diagonal
    // two-level PCG (pcg_multi.hip: pcg2_solve) for reduced systems no band factor applies to: aggregates of consecutive free
    // cameras x six rigid-body modes (ba_coarse_basis_kernel)
    bool two_level = false;
    uint32_t tl_n_agg = 0, tl_ncp = 0, tl_n_cb = 0;
    int tl_last_it = 0;
    DevBuf<uint32_t> tl_agg_ptr, tl_row_agg, tl_agg_ref, tl_cb_ptr, tl_cb_ent, tl_cb_I, tl_cb_J;
    DevBuf<double> tl_P, tl_G, tl_Ac0, tl_Ainv, tl_ebuf, tl_rc, tl_status;
    int cr_rounds = 1;                  // PCG rounds enqueued per solve with the exact band factor (see take_step)
    bool cr_factor_valid = false;       // cr_ws_of(fac_idx) holds a complete factor of an earlier (or this) iteration's reduced matrix
    // lagged factor (take_step): two workspaces, the refresh on a second stream
    DevBuf<double> cr_ws2, lag_status;
    hipStream_t fstream = nullptr;
    hipEvent_t ev_s_ready = nullptr, ev_factor = nullptr;
    int fac_idx = 0;                    // workspace of the newest complete factor
    bool fac_pending = false;           // a refresh into the other workspace is in flight (ev_factor)
    bool lag_ok = true;                 // the last lagged solve stayed within its rounds
    bool lagged_solve = false;          // this iteration's solve used the lagged factor
    int lag_rounds = 3;                 // PCG rounds enqueued with a lagged factor
    double last_rel_decrease = 1.0;     // (cost - candidate) / cost of the last accepted step
    double* cr_ws_of(int i) { return i == 0 ? cr_ws.p : cr_ws2.p; }
    int lag_init()
    {
        if (fstream) return SOSLAM_OK;
        SOSLAM_HIP_CHECK(hipStreamCreateWithFlags(&fstream, hipStreamNonBlocking));
        SOSLAM_HIP_CHECK(hipEventCreateWithFlags(&ev_s_ready, hipEventDisableTiming));
        SOSLAM_HIP_CHECK(hipEventCreateWithFlags(&ev_factor, hipEventDisableTiming));
        return SOSLAM_OK;
    }
    int solver = SOSLAM_SOLVER_PCG;
    double setup_seconds = 0.0;

    // host-side permutations (internal <-> caller order)
    std::vector<uint32_t> pt_int2user, pt_user2int, obs_int2user;
    // host scratch of build_problem that is as long as the observation list: kept with the handle, so that a second problem of
    // the same size pays neither the allocation nor the zero fill of 40 MB of vectors (4 ms at 1 M observations)
    std::vector<float4> hs_uv;
    std::vector<uint32_t> hs_obs_pt, hs_obs_cam, hs_pt_obs, hs_q_pt, hs_q_cam;
    std::vector<int32_t> h_cam_free;
    std::vector<uint32_t> h_blk_row, h_blk_col;
    std::vector<std::pair<uint32_t, uint32_t>> covis;   // job-wide camera pairs (multi-GPU)

    // static device data
    DevBuf<float4> uv;
    DevBuf<uint32_t> obs_pt, cam_tile_start, pt_start, pt_obs, q_pt, q_cam;
    DevBuf<uint8_t> q_slot, ent_trans;
    DevBuf<Tile> tiles;
    DevBuf<int32_t> cam_free, diag_block;
    DevBuf<uint32_t> chunk_slab, blk_contrib_ptr, blk_contrib_off, cam_contrib_ptr, cam_contrib_off;
    DevBuf<double> slab;
    DevBuf<SchurChunk> chunks;
    DevBuf<SchurBatch> batches;
    DevBuf<uint32_t> pair_row;
    // long-track points (beyond the Schur window / batch limits)
    DevBuf<LongPoint> long_pts;
    DevBuf<uint32_t> lo_row, lo_cam, lo_cam_off, pair_a, pair_b, pair_off;
    DevBuf<uint32_t> chunk_cam, free_cam;   // camera of each Schur window slot / of each free index
    DevBuf<double> long_wy;
    uint32_t n_long = 0, n_long_pairs = 0, n_short = 0;
    DevBuf<uint32_t> row_ptr, ent_col, ent_blk, blk_row, blk_col;
    DevBuf<int32_t> cr_map;             // gather map of the cyclic-reduction assembly (crsolve.hip)
    DevBuf<uint32_t> cr_comp_ptr, cr_comp_ent;   // off-band mode: per camera the blocks the map leaves out
    DevBuf<double> cr_comp;             // their absolute row sums (diagonal compensation of the factored band)

    // state and work buffers
    DevBuf<double> cams[2], pts[2];
    int cur = 0;
    DevBuf<double> cam_part;            // [cam_update_blocks][5] partials of the camera share of the step scalars
    DevBuf<double> campre, campre_c, ar, dcw, tile_part, cost_part, C, gp, sp, Cinv, ptfac, sc, lc, dc_free, dc_full, dp, part;
    DevBuf<double> lin_resid, lin_work, dense, band, bandT, band_dinv, cr_ws;
    DevBuf<double> reduce_own;          // library-owned reduce buffer
    double* reduce = nullptr;           // [S blocks | rhs | diagB | gc_red | tail(4)] [scalars(SC_COUNT)]
    uint64_t reduce_main = 0;           // f64 in the per-iteration system payload
    uint64_t reduce_count = 0;          // reduce_main + SC_COUNT: everything an all-reduce may touch
    double* host_raw = nullptr;         // pinned, 4 + SC_COUNT: the tail and the scalars as they lie in the reduce buffer
    double* host_scal = nullptr;        // host_raw + 4
    unsigned long long* host_seq = nullptr;   // behind host_raw: sequence number of the last publication
    unsigned long long publish_seq = 0;
    unsigned char* up_host[3] = {nullptr, nullptr, nullptr};   // pinned staging of the packed transfers (PackedSeg table, then payloads):
    size_t up_host_bytes[3] = {0, 0, 0};                       // the index arrays; the table of the fills (the first kernel may still be
                                                               // reading); the state, in and out (set_state / get_state)
    // One pinned allocation serves the small buffers above and below (hipHostMalloc costs 0.15 - 0.25 ms a call: five of them were
    // most of a one-shot per-frame call); what outgrows it gets an allocation of its own
    unsigned char* pin_pool = nullptr;
    size_t pin_pool_bytes = 0, pin_pool_used = 0;
    int pinned_alloc(void** out, size_t bytes)
    {
        constexpr size_t kPool = 512u << 10;
        if (!pin_pool) {
            if (hipHostMalloc(reinterpret_cast<void**>(&pin_pool), kPool, hipHostMallocDefault) != hipSuccess) { pin_pool = nullptr; }
            else { pin_pool_bytes = kPool; pin_pool_used = 0; }
        }
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (pin_pool && pin_pool_used + need <= pin_pool_bytes) {
            *out = pin_pool + pin_pool_used;
            pin_pool_used += need;
            return SOSLAM_OK;
        }
        SOSLAM_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
        return SOSLAM_OK;
    }
    void pinned_free(void* q)   // (a piece of the pool stays with the pool)
    {
        if (!q) return;
        const unsigned char* u = static_cast<const unsigned char*>(q);
        if (pin_pool && u >= pin_pool && u < pin_pool + pin_pool_bytes) return;
        (void)hipHostFree(q);
    }
    double* ps_host = nullptr;          // pinned: record (PSV_COUNT) + sequence word + iteration log of the resident structure-only solve
    int ps_host_entries = 0;
    DevBuf<double> ps_log;              // the same log on the device while the kernel runs
    DevBuf<unsigned int> arrivals;      // ba_apply_small: arrival counter of its workgroups (zero between launches)
    DevBuf<double> ps_sync;             // its grid barrier: [2][kPointsOnlyMax / 64][16] partial-sum records, then the arrival counter
    unsigned long long ps_sync_base = 0;   // value of that counter between launches

    // multi-GPU
    soslam_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    soslam_host_allreduce_fn host_allreduce = nullptr;   // collective on host memory: the library stages through `stage`
    void* host_allreduce_user = nullptr;
    double* stage = nullptr;            // pinned staging buffer of the host collective
    uint64_t stage_count = 0;
    unsigned long long* stage_seq = nullptr;   // behind it: sequence number of the last download
    unsigned long long stage_seq_next = 0;
    RcclComm* rccl = nullptr;           // the library's own RCCL communicator (soslam_ba_init_rccl); wins over the callback
    int rank = 0, world = 1;
    // a collective is attached (the RCCL leg or a callback): the iteration takes the multi-rank path - payload and scalar
    // all-reduces, acceptance test after them - whatever the number of ranks (a one-rank job exercises every step of it)
    bool collective() const { return rccl != nullptr || allreduce != nullptr || host_allreduce != nullptr; }
    bool stop_agreed = false;           // multi-rank: some rank voted to end the solve (time limit) in the last iteration
    DevBuf<double> gather;              // soslam_ba_get_state_global: all ranks' points
    DevBuf<double> agree;               // soslam_ba_agree_status: one word
    DevBuf<double> ls_tile, ls_part;    // line-search trial: per-tile {rho, direction . gradient}, per-block {|step|^2, max |delta|}

    // trust region
    double radius = 0.0, decrease_factor = 2.0, x_cost = 0.0;
    bool linearized = false, scale_init = false;
    bool pt_blocks_valid = false;       // C / gp belong to the compact rows held (else the ten-camera Schur kernel forms them itself)
    bool x_cost_known = false;          // x_cost is the cost at cams[cur], pts[cur] (accepted candidates: no sum over the tiles needed)
    bool campre_current = false;        // campre already holds the pose table of cams[cur] (set on acceptance, used once by linearize)
    bool points_only_ready = false;     // structure-only path: both pose tables / camera buffers hold the constant poses, dc_full is zero
    bool points_resident_ready = false; // ... the resident solve's share of that: the pose table of the constant cameras
    int invalid_run = 0;
    std::vector<soslam_ba_iteration> log;
    std::vector<std::vector<uint32_t>> h_blk_contrib, h_cam_contrib;   // set-up scratch (build_problem), kept for its capacity

    // profiling
    std::vector<hipEvent_t> ev_pool;
    struct StageMark { int stage; size_t e0, e1; };
    std::vector<StageMark> marks;
    size_t ev_used = 0;

    CompactRows rows() const { return CompactRows{ar.p, ar.p + (size_t)kArG * n_obs}; }   // G rows, then the h rows (ba_kernels.h)
    double* S() const { return reduce; }
    double* rhs() const { return reduce + (size_t)n_blocks * 36; }
    double* diagB() const { return rhs() + (size_t)n_free * 6; }
    double* gc_red() const { return diagB() + (size_t)n_free * 6; }
    double* tail() const { return gc_red() + (size_t)n_free * 6; }
    double* scalp() const { return reduce + reduce_main; }

    ~soslam_ba()
    {
        for (auto e : ev_pool) (void)hipEventDestroy(e);
        pinned_free(host_raw);
        pinned_free(ps_host);
        for (unsigned char* u : up_host) pinned_free(u);
        if (pin_pool) (void)hipHostFree(pin_pool);
        if (stage) (void)hipHostFree(stage);
        rccl_comm_destroy(rccl);
        if (fstream) { (void)hipStreamSynchronize(fstream); (void)hipStreamDestroy(fstream); }
        if (ev_s_ready) (void)hipEventDestroy(ev_s_ready);
        if (ev_factor) (void)hipEventDestroy(ev_factor);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

LmDiag lm_diag(const soslam_ba* h, double radius) { return LmDiag{radius, h->opt.min_lm_diagonal, h->opt.max_lm_diagonal}; }

// Ceres runs TrustRegionMinimizer::DoLineSearch only when the problem is constrained (options.is_constrained: some parameter has
// a finite bound).  The reference bounds every point coordinate (/root/reference/src/bundle_adjuster.cpp:104-108), so its problems
// always are; a C-ABI caller with infinite bounds gets plain trust-region steps.
bool is_constrained(const soslam_ba* h) { return std::isfinite(h->opt.lower_bound) || std::isfinite(h->opt.upper_bound); }

// ---- profiling marks ---------------------------------------------------------------------------------

struct StageScope {
    soslam_ba* h;
    int stage;
    size_t idx = 0;
    bool on;
    StageScope(soslam_ba* h_, int stage_) : h(h_), stage(stage_), on(h_->opt.profile_stages != 0)
    {
        if (!on) return;
        while (h->ev_pool.size() < h->ev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
            h->ev_pool.push_back(e);
        }
        idx = h->ev_used;
        h->ev_used += 2;
        (void)hipEventRecord(h->ev_pool[idx], h->stream);
    }
    ~StageScope()
    {
        if (!on) return;
        (void)hipEventRecord(h->ev_pool[idx + 1], h->stream);
        h->marks.push_back({stage, idx, idx + 1});
    }
};

void collect_stage_times(soslam_ba* h, soslam_ba_summary* s)
{
    if (!h->opt.profile_stages || !s) { h->marks.clear(); h->ev_used = 0; return; }
    (void)hipStreamSynchronize(h->stream);
    for (const auto& m : h->marks) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, h->ev_pool[m.e0], h->ev_pool[m.e1]) == hipSuccess) {
            s->stage_ms[m.stage] += ms;
            s->stage_calls[m.stage]++;
        }
    }
    h->marks.clear();
    h->ev_used = 0;
}

// ---- index construction ------------------------------------------------------------------------------

// SOSLAM_SETUP_TIMING=1 prints where set_problem spends its host time (development aid)
#define SETUP_MARK(label)                                                                        \
    do {                                                                                         \
        if (setup_timing) {                                                                      \
            const double t_now = now_sec();                                                      \
            std::fprintf(stderr, "[setup] %-14s %8.3f ms\n", label, 1e3 * (t_now - t_mark));     \
            t_mark = t_now;                                                                      \
        }                                                                                        \
    } while (0)

// Host threads for the memory-bound passes of build_problem: [0, n) in contiguous ranges, one per thread, when the pass is
// large enough to pay for starting them (the reference's windows never are; a 1 M-observation problem is).  Every range
// writes its own part of the output arrays, so the result does not depend on the number of threads.
unsigned parallel_threads(size_t n, size_t work)
{
    const unsigned nt = work < 200000 ? 1u : std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    return n < nt ? 1u : nt;
}

template <class F>
void parallel_ranges(size_t n, size_t work, F&& f)
{
    const unsigned nt = parallel_threads(n, work);
    if (nt <= 1) { f((size_t)0, n, 0u); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const size_t lo = (size_t)t * per, hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back([&f, lo, hi, t] { f(lo, hi, t); });
    }
    for (auto& x : th) x.join();
}

// Uploads of one set_problem, collected: up to kPackedUploadMax bytes go as ONE kernel reading a pinned staging buffer
// (launch_packed_scatter), more as one copy command per array.  The staging buffer is free again when the stream has been
// synchronised (build_problem ends on that).
constexpr size_t kPackedUploadMax = 4u << 20;
struct UploadBatch {
    struct Seg { void* dst; const void* src; size_t bytes; };
    std::vector<Seg> segs;
    size_t total = 0;
    template <class T>
    int add(DevBuf<T>& b, const std::vector<T>& v)
    {
        SOSLAM_CHECK(b.alloc(v.size()));
        if (!v.empty()) { segs.push_back(Seg{b.p, v.data(), v.size() * sizeof(T)}); total += (v.size() * sizeof(T) + 15) & ~(size_t)15; }
        return SOSLAM_OK;
    }
    template <class T>
    int add_zero(DevBuf<T>& b)
    {
        if (b.n) segs.push_back(Seg{b.p, nullptr, b.n * sizeof(T)});
        return SOSLAM_OK;
    }
    int flush(soslam_ba* h, hipStream_t s, int region)
    {
        static const bool off = [] { const char* e = getenv("SOSLAM_NO_PACKED_UPLOAD"); return e && *e && *e != '0'; }();
        if (segs.empty()) return SOSLAM_OK;
        if (off || total > kPackedUploadMax) {
            for (const Seg& g : segs) {
                if (g.src) SOSLAM_HIP_CHECK(hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyHostToDevice, s));
                else SOSLAM_HIP_CHECK(hipMemsetAsync(g.dst, 0, g.bytes, s));
            }
            segs.clear(); total = 0;
            return SOSLAM_OK;
        }
        const size_t table_bytes = (segs.size() * sizeof(PackedSeg) + 15) & ~(size_t)15, need = table_bytes + total;
        if (need > h->up_host_bytes[region]) {
            h->pinned_free(h->up_host[region]);
            h->up_host[region] = nullptr; h->up_host_bytes[region] = 0;
            const size_t cap = std::max<size_t>(need + need / 4, region == 1 ? (size_t)4096 : (size_t)(1u << 16));
            SOSLAM_CHECK(h->pinned_alloc(reinterpret_cast<void**>(&h->up_host[region]), cap));
            h->up_host_bytes[region] = cap;
        }
        PackedSeg* table = reinterpret_cast<PackedSeg*>(h->up_host[region]);
        unsigned char* payload = h->up_host[region] + table_bytes;
        size_t off_b = 0;
        int n_seg = 0;
        for (const Seg& g : segs) {
            if (g.src) {
                std::memcpy(payload + off_b, g.src, g.bytes);
                table[n_seg++] = PackedSeg{g.dst, (uint64_t)off_b, (uint64_t)g.bytes};
                off_b += (g.bytes + 15) & ~(size_t)15;
            } else if (g.bytes > (1u << 20)) {
                SOSLAM_HIP_CHECK(hipMemsetAsync(g.dst, 0, g.bytes, s));   // a large fill is the copy engine's
            } else {
                table[n_seg++] = PackedSeg{g.dst, ~0ull, (uint64_t)g.bytes};
            }
        }
        launch_packed_scatter(s, table, n_seg, payload);
        SOSLAM_HIP_CHECK(hipGetLastError());
        segs.clear(); total = 0;
        return SOSLAM_OK;
    }
};

int build_problem(soslam_ba* h, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs, const uint32_t* ocam,
                  const uint32_t* opt_, const float* ouv, const uint8_t* fixed)
{
    if (h->fstream) SOSLAM_HIP_CHECK(hipStreamSynchronize(h->fstream));   // a refresh of the lagged factor may still read S and write a workspace
    h->fac_pending = false;

    const double t0 = now_sec();
    const bool setup_timing = std::getenv("SOSLAM_SETUP_TIMING") != nullptr;
    double t_mark = t0;
    hipStream_t s = h->stream;
    for (uint32_t k = 0; k < n_obs; k++)
        if (ocam[k] >= n_cam || opt_[k] >= n_pt) {
            set_last_error("observation %u refers to camera %u / point %u out of range", k, ocam[k], opt_[k]);
            return SOSLAM_ERR_INVALID_ARGUMENT;
        }
    h->n_cam = n_cam; h->n_pt = n_pt; h->n_obs = n_obs;

    // free-camera numbering
    h->h_cam_free.assign(n_cam, -1);
    uint32_t nf = 0;
    std::vector<uint32_t> free_cam;   // camera of each free index
    for (uint32_t c = 0; c < n_cam; c++)
        if (!(fixed && fixed[c])) { h->h_cam_free[c] = (int32_t)nf++; free_cam.push_back(c); }
    h->n_free = nf;

    SETUP_MARK("begin");
    // internal point order: by (first camera, last camera, caller id) so that neighbouring points share
    // their camera window (the Schur kernel's chunks) and a camera bucket reads near-contiguous points
    std::vector<uint32_t> pmin(n_pt, UINT32_MAX), pmax(n_pt, 0), pcnt(n_pt, 0), pfree(n_pt, 0);
    for (uint32_t k = 0; k < n_obs; k++) {
        const uint32_t p = opt_[k], c = ocam[k];
        pmin[p] = std::min(pmin[p], c); pmax[p] = std::max(pmax[p], c); pcnt[p]++;
        if (h->h_cam_free[c] >= 0) pfree[p]++;
    }
    // long tracks (more free cameras than the widest Schur window, or more observations than a batch) go last
    // and are eliminated by the long-track kernels
    constexpr uint32_t kWindowMax = 32;
    std::vector<uint8_t> is_long(n_pt, 0);
    uint32_t n_long = 0;
    for (uint32_t p = 0; p < n_pt; p++)
        if (pfree[p] > kWindowMax || pcnt[p] > (uint32_t)kBatchObs) { is_long[p] = 1; n_long++; }
    const uint32_t n_short = n_pt - n_long;
    // order (long flag, first camera, last camera, caller id): two stable counting sorts over the cameras, then a
    // stable partition - linear in the points, no comparator calls
    h->pt_int2user.resize(n_pt);
    {
        std::vector<uint32_t> tmp(n_pt), cnt(n_cam + 2);
        auto counting_pass = [&](const std::vector<uint32_t>& key, const uint32_t* in, uint32_t* out, bool identity_in) {
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (uint32_t i = 0; i < n_pt; i++) cnt[std::min(key[identity_in ? i : in[i]], n_cam) + 1]++;   // unobserved points: key n_cam
            for (uint32_t c = 0; c <= n_cam; c++) cnt[c + 1] += cnt[c];
            for (uint32_t i = 0; i < n_pt; i++) {
                const uint32_t id = identity_in ? i : in[i];
                out[cnt[std::min(key[id], n_cam)]++] = id;
            }
        };
        counting_pass(pmax, nullptr, tmp.data(), true);                        // least significant key first
        counting_pass(pmin, tmp.data(), h->pt_int2user.data(), false);
        std::stable_partition(h->pt_int2user.begin(), h->pt_int2user.end(), [&](uint32_t id) { return !is_long[id]; });
    }
    h->pt_user2int.resize(n_pt);
    for (uint32_t i = 0; i < n_pt; i++) h->pt_user2int[h->pt_int2user[i]] = i;

    SETUP_MARK("point order");
    // camera-major observation order: (camera, internal point)
    // (camera, internal point, caller index): stable counting sort by internal point, then by camera
    h->obs_int2user.resize(n_obs);
    // The reference hands observations over frame by frame (bundle_adjuster.cpp:62-74), points in first-seen order: such an
    // input is already in (camera, internal point, caller index) order.  One linear check instead of two counting sorts
    // over the observations; the order itself, and every bit of the results, are those of the sort.
    bool presorted = true;
    {
        uint8_t bad[16] = {0};
        parallel_ranges(n_obs, n_obs, [&](size_t lo, size_t hi, unsigned t) {
            bool ok = true;
            for (size_t k = std::max<size_t>(lo, 1); k < hi && ok; k++)
                ok = ocam[k] > ocam[k - 1] || (ocam[k] == ocam[k - 1] && h->pt_user2int[opt_[k]] >= h->pt_user2int[opt_[k - 1]]);
            if (!ok) bad[t] = 1;
        });
        for (uint8_t b1 : bad) presorted = presorted && !b1;
    }
    if (presorted) {
        std::iota(h->obs_int2user.begin(), h->obs_int2user.end(), 0u);
    } else {
        std::vector<uint32_t> tmp(n_obs), cnt(std::max(n_pt, n_cam) + 1);
        std::fill(cnt.begin(), cnt.begin() + n_pt + 1, 0u);
        for (uint32_t k = 0; k < n_obs; k++) cnt[h->pt_user2int[opt_[k]] + 1]++;
        for (uint32_t p = 0; p < n_pt; p++) cnt[p + 1] += cnt[p];
        for (uint32_t k = 0; k < n_obs; k++) tmp[cnt[h->pt_user2int[opt_[k]]]++] = k;
        std::fill(cnt.begin(), cnt.begin() + n_cam + 1, 0u);
        for (uint32_t k = 0; k < n_obs; k++) cnt[ocam[k] + 1]++;
        for (uint32_t c = 0; c < n_cam; c++) cnt[c + 1] += cnt[c];
        for (uint32_t i = 0; i < n_obs; i++) h->obs_int2user[cnt[ocam[tmp[i]]]++] = tmp[i];
    }
    std::vector<float4>& uv = h->hs_uv;
    std::vector<uint32_t>&v_obs_pt = h->hs_obs_pt, &v_obs_cam = h->hs_obs_cam;
    uv.resize(n_obs); v_obs_pt.resize(n_obs); v_obs_cam.resize(n_obs);   // every element is written below
    std::vector<uint32_t> cam_start(n_cam + 1, 0);
    {
        std::vector<std::vector<uint32_t>> hist(16);
        parallel_ranges(n_obs, n_obs, [&](size_t lo, size_t hi, unsigned t) {
            std::vector<uint32_t>& hc = hist[t];
            hc.assign(n_cam + 1, 0);
            for (size_t i = lo; i < hi; i++) {
                const uint32_t k = h->obs_int2user[i];
                uv[i] = make_float4(ouv[4 * (size_t)k], ouv[4 * (size_t)k + 1], ouv[4 * (size_t)k + 2], ouv[4 * (size_t)k + 3]);
                v_obs_pt[i] = h->pt_user2int[opt_[k]];
                v_obs_cam[i] = ocam[k];
                hc[ocam[k] + 1]++;
            }
        });
        for (const auto& hc : hist)
            for (size_t c = 0; c < hc.size(); c++) cam_start[c] += hc[c];
    }
    for (uint32_t c = 0; c < n_cam; c++) cam_start[c + 1] += cam_start[c];

    SETUP_MARK("obs order");
    // tiles: <= kTileObs observations of one camera each
    std::vector<Tile> tiles;
    std::vector<uint32_t> cam_tile_start(n_cam + 1, 0);
    // A camera's observations are dealt evenly over its tiles, in whole waves (2 000 observations: 1024 + 976).  Smaller
    // tiles (SOSLAM_TILE_OBS, development) were measured slower at 1 M observations: 26.3 us at 1024, 28.5 at 512, 35.0 at 256 -
    // the per-tile epilogue outweighs the better balance of the last workgroups.
    uint32_t tile_obs = kTileObs;
    if (const char* e = std::getenv("SOSLAM_TILE_OBS")) tile_obs = (uint32_t)std::min<long>(std::max<long>(std::atol(e), 64), kTileObs);
    for (uint32_t c = 0; c < n_cam; c++) {
        cam_tile_start[c] = (uint32_t)tiles.size();
        const uint32_t cnt = cam_start[c + 1] - cam_start[c];
        if (!cnt) continue;
        const uint32_t nt = div_up(cnt, tile_obs);
        const uint32_t per = std::min<uint32_t>(div_up(div_up(cnt, nt), 64u) * 64u, kTileObs);
        for (uint32_t b = cam_start[c]; b < cam_start[c + 1]; b += per)
            tiles.push_back(Tile{c, b, std::min<uint32_t>(per, cam_start[c + 1] - b), 0});
    }
    cam_tile_start[n_cam] = (uint32_t)tiles.size();
    h->n_tiles = (uint32_t)tiles.size();

    // point-major lists (camera ascending inside a point because the scan is camera-major)
    std::vector<uint32_t> pt_start(n_pt + 1, 0);
    std::vector<uint32_t>&pt_obs = h->hs_pt_obs, &q_pt = h->hs_q_pt, &q_cam = h->hs_q_cam;
    pt_obs.resize(n_obs); q_pt.resize(n_obs); q_cam.resize(n_obs);        // every element is written below
    for (uint32_t i = 0; i < n_obs; i++) pt_start[v_obs_pt[i] + 1]++;
    for (uint32_t p = 0; p < n_pt; p++) pt_start[p + 1] += pt_start[p];
    {
        // (a thread-parallel form of this counting sort - per-thread counts of 100 k points, then scatter - was measured at
        // 1 M observations: 6.9 ms against 3.1 ms sequential; the counts do not fit the cores' caches)
        std::vector<uint32_t> fill(pt_start.begin(), pt_start.end() - 1);
        for (uint32_t i = 0; i < n_obs; i++) {
            const uint32_t q = fill[v_obs_pt[i]]++;
            pt_obs[q] = i;
            q_pt[q] = v_obs_pt[i];
            q_cam[q] = v_obs_cam[i];
        }
    }

    SETUP_MARK("tiles+lists");
    // block-sparse pattern of the reduced camera matrix: pairs of free cameras that share a point
    std::vector<std::vector<uint32_t>> rows(nf);
    uint32_t max_track = 0;
    {
        // up to 4096 free cameras the pairs are marked in an upper-triangular bitmap (2 MB at most) and the rows read
        // off it; beyond that they are collected per row and sorted
        const bool use_bitmap = nf <= 4096;
        const size_t words_per_row = (nf + 63) / 64;
        std::vector<uint64_t> bits(use_bitmap ? (size_t)nf * words_per_row : 0, 0);
        auto mark = [&](uint32_t a, uint32_t b) {   // a <= b
            if (use_bitmap) bits[(size_t)a * words_per_row + (b >> 6)] |= 1ull << (b & 63);
            else rows[a].push_back(b);
        };
        std::vector<uint32_t> fc, fc_prev;
        for (uint32_t f = 0; f < nf; f++) mark(f, f);
        for (uint32_t p = 0; p < n_pt; p++) {
            fc.clear();
            for (uint32_t q = pt_start[p]; q < pt_start[p + 1]; q++) {
                const int32_t f = h->h_cam_free[q_cam[q]];
                if (f < 0) continue;
                // A (camera, point) pair observed twice would own one window slot and one row of the pair table: the second row
                // would be dropped from W while C, g_p, B and g_c (plain sums over the rows) kept it - an inconsistent reduced
                // system.  Rejected here as it is for long-track points below (a point's observations are camera-ascending, so
                // duplicates are neighbours); the reference's front end never produces them (one match per frame and point).
                if (!fc.empty() && fc.back() == (uint32_t)f) {
                    set_last_error("point %u is observed twice by camera %u", h->pt_int2user[p], q_cam[q]);
                    return SOSLAM_ERR_INVALID_ARGUMENT;
                }
                fc.push_back((uint32_t)f);
            }
            if (p < n_short) max_track = std::max<uint32_t>(max_track, (uint32_t)fc.size());
            if (fc == fc_prev) continue;   // neighbouring points mostly share their camera set: its pairs are marked already
            for (size_t a = 0; a < fc.size(); a++)
                for (size_t b = a + 1; b < fc.size(); b++) mark(fc[a], fc[b]);
            fc_prev.swap(fc);
        }
        for (const auto& pr : h->covis) {
            if (pr.first >= n_cam || pr.second >= n_cam) {
                set_last_error("covisibility pair (%u, %u) out of range", pr.first, pr.second);
                return SOSLAM_ERR_INVALID_ARGUMENT;
            }
            const int32_t fa = h->h_cam_free[pr.first], fb = h->h_cam_free[pr.second];
            if (fa < 0 || fb < 0) continue;
            mark((uint32_t)std::min(fa, fb), (uint32_t)std::max(fa, fb));
        }
        if (use_bitmap) {
            for (uint32_t f = 0; f < nf; f++)
                for (size_t w = f >> 6; w < words_per_row; w++) {
                    uint64_t m = bits[(size_t)f * words_per_row + w];
                    while (m) {
                        rows[f].push_back((uint32_t)(w * 64 + (size_t)__builtin_ctzll(m)));
                        m &= m - 1;
                    }
                }
        } else {
            for (auto& r : rows) {
                std::sort(r.begin(), r.end());
                r.erase(std::unique(r.begin(), r.end()), r.end());
            }
        }
    }
    // window width of the Schur kernel: the narrowest instantiation that holds the longest windowed track (10 cameras =
    // 60 rows fill four 16-row tiles of the matrix cores exactly; 11 would need a fifth tile row: 15 tiles instead of 10)
    h->kmax = max_track > 20 ? 32 : (max_track > 16 ? 20 : (max_track > 10 ? 16 : 10));
    std::vector<uint32_t> row_first(nf + 1, 0);
    for (uint32_t f = 0; f < nf; f++) row_first[f + 1] = row_first[f] + (uint32_t)rows[f].size();
    h->n_blocks = row_first[nf];
    h->h_blk_row.resize(h->n_blocks); h->h_blk_col.resize(h->n_blocks);
    std::vector<int32_t> diag_block(nf);
    for (uint32_t f = 0; f < nf; f++)
        for (size_t e = 0; e < rows[f].size(); e++) {
            const uint32_t b = row_first[f] + (uint32_t)e;
            h->h_blk_row[b] = f; h->h_blk_col[b] = rows[f][e];
            if (rows[f][e] == f) diag_block[f] = (int32_t)b;
        }
    auto find_block = [&](uint32_t i, uint32_t j) -> int32_t {
        const auto& r = rows[i];
        auto it = std::lower_bound(r.begin(), r.end(), j);
        if (it == r.end() || *it != j) return -1;
        return (int32_t)(row_first[i] + (uint32_t)(it - r.begin()));
    };
    // full row lists (both triangles) for the matrix-vector product
    std::vector<uint32_t> row_ptr(nf + 1, 0), ent_col, ent_blk;
    std::vector<uint8_t> ent_trans;
    {
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> full(nf);  // (col, blk | trans<<31)
        for (uint32_t b = 0; b < h->n_blocks; b++) {
            const uint32_t i = h->h_blk_row[b], j = h->h_blk_col[b];
            full[i].push_back({j, b});
            if (i != j) full[j].push_back({i, b | 0x80000000u});
        }
        for (uint32_t f = 0; f < nf; f++) {
            std::sort(full[f].begin(), full[f].end());
            row_ptr[f + 1] = row_ptr[f] + (uint32_t)full[f].size();
            for (auto& e : full[f]) {
                ent_col.push_back(e.first);
                ent_blk.push_back(e.second & 0x7FFFFFFFu);
                ent_trans.push_back((uint8_t)(e.second >> 31));
            }
        }
    }

    SETUP_MARK("pattern");
    // Schur chunks: consecutive points whose free cameras fit kmax local slots; batches of <= 128 obs
    const int K = h->kmax;
    // points per chunk: about two chunks per CU (a chunk's window goes to its own slab, so finer chunks cost slab traffic in
    // ba_schur_reduce - measured at config 3: 204-point chunks 85 us for the stage, 68-point chunks 92 us although the
    // kernel alone is faster); narrow windows close at every change of the camera set anyway
    uint32_t chunk_pts_max = std::max<uint32_t>(32, std::min<uint32_t>(1024, n_pt / (h->kmax <= 10 ? 448 : 512) + 1));
    if (const char* e = std::getenv("SOSLAM_CHUNK_PTS")) chunk_pts_max = (uint32_t)std::max(1, std::atoi(e));   // development
    std::vector<SchurChunk> chunks;
    std::vector<SchurBatch> batches;
    std::vector<uint32_t> pair_row;                         // kmax <= 10: [batch][120], see launch_schur
    std::vector<uint32_t> chunk_slab;                       // offset of each chunk's window in the slab
    std::vector<uint32_t> chunk_cam;                        // [chunk][K] camera of each window slot
    // (the lists of lists stay with the handle: a kept handle's next window re-uses their storage)
    std::vector<std::vector<uint32_t>>& blk_contrib = h->h_blk_contrib;
    std::vector<std::vector<uint32_t>>& cam_contrib = h->h_cam_contrib;
    if (blk_contrib.size() < h->n_blocks) blk_contrib.resize(h->n_blocks);
    if (cam_contrib.size() < nf) cam_contrib.resize(nf);
    for (uint32_t b = 0; b < h->n_blocks; b++) blk_contrib[b].clear();
    for (uint32_t f = 0; f < nf; f++) cam_contrib[f].clear();
    uint64_t slab_count = 0;
    std::vector<uint8_t> q_slot(n_obs, 255);
    std::vector<uint32_t> chunk_p_range, chunk_local;       // [chunk][2] point range, [chunk][K] free index of each window slot
    {
        std::vector<uint32_t> local, merged, fc;
        uint32_t chunk_p0 = 0;
        auto close_chunk = [&](uint32_t p_end) {
            if (p_end == chunk_p0) return;
            SchurChunk ch{};
            ch.batch_begin = (uint32_t)batches.size();
            ch.n_local = (uint32_t)local.size();
            uint32_t bp = chunk_p0;
            while (bp < p_end) {
                uint32_t be = bp, nq = 0;
                while (be < p_end && (be - bp) < (uint32_t)schur_batch_points(K) && nq + (pt_start[be + 1] - pt_start[be]) <= (uint32_t)kBatchObs) {
                    nq += pt_start[be + 1] - pt_start[be];
                    be++;
                }
                if (be == bp) be = bp + 1;  // single point wider than a batch cannot happen (track <= 32 + fixed)
                batches.push_back(SchurBatch{pt_start[bp], pt_start[be], bp, be, 0u});
                bp = be;
            }
            ch.batch_end = (uint32_t)batches.size();
            // the per-observation part (window slots, the ten-camera kernel's pair table, the batches' "full" flags) is done
            // for all chunks together after this loop, chunk-parallel: it only needs the chunk's point range and cameras
            chunk_p_range.push_back(chunk_p0);
            chunk_p_range.push_back(p_end);
            for (int a = 0; a < K; a++) chunk_local.push_back(a < (int)local.size() ? local[a] : UINT32_MAX);
            // slab layout of this chunk: [pair (a <= b < n_local)][36] then [camera a][6]; every pair whose block
            // exists in the pattern (it always does: the pattern is a superset) feeds that block's list
            const uint32_t base = (uint32_t)slab_count;
            chunk_slab.push_back(base);
            const int KL = (int)local.size();
            for (int a = 0; a < K; a++) chunk_cam.push_back(a < KL ? free_cam[local[a]] : 0u);   // camera of each window slot
            int pair = 0;
            for (int a = 0; a < KL; a++)
                for (int b = a; b < KL; b++, pair++) {
                    const int32_t blk = find_block(local[a], local[b]);
                    if (blk >= 0) blk_contrib[(size_t)blk].push_back(base + (uint32_t)pair * 36);
                }
            for (int a = 0; a < KL; a++) cam_contrib[local[a]].push_back(base + (uint32_t)pair * 36 + (uint32_t)a * 6);
            slab_count += (uint64_t)pair * 36 + (uint64_t)KL * 6;
            chunks.push_back(ch);
            chunk_p0 = p_end;
            local.clear();
        };
        std::vector<uint32_t> fc_last;
        for (uint32_t p = 0; p < n_short; p++) {
            fc.clear();
            for (uint32_t q = pt_start[p]; q < pt_start[p + 1]; q++) {
                const int32_t f = h->h_cam_free[q_cam[q]];
                if (f < 0) continue;
                // A (camera, point) pair observed twice would own one window slot and one row of the pair table: the second row
                // would be dropped from W while C, g_p, B and g_c (plain sums over the rows) kept it - an inconsistent reduced
                // system.  Rejected here as it is for long-track points below (a point's observations are camera-ascending, so
                // duplicates are neighbours); the reference's front end never produces them (one match per frame and point).
                if (!fc.empty() && fc.back() == (uint32_t)f) {
                    set_last_error("point %u is observed twice by camera %u", h->pt_int2user[p], q_cam[q]);
                    return SOSLAM_ERR_INVALID_ARGUMENT;
                }
                fc.push_back((uint32_t)f);
            }
            if (p > chunk_p0 && p - chunk_p0 < chunk_pts_max && fc == fc_last) continue;   // same camera set as the point before: the window does not change
            merged.clear();
            std::set_union(local.begin(), local.end(), fc.begin(), fc.end(), std::back_inserter(merged));
            if ((int)merged.size() > K || p - chunk_p0 >= chunk_pts_max) {
                close_chunk(p);
                merged = fc;
            }
            local.swap(merged);
            fc_last = fc;
        }
        close_chunk(n_short);
    }
    SETUP_MARK("chunk windows");
    if (K <= 10) pair_row.assign(batches.size() * (size_t)kS10PairsPerBatch, 0xFFFFFFFFu);
    parallel_ranges(chunks.size(), n_obs, [&](size_t c_lo, size_t c_hi, unsigned) {
        for (size_t ci = c_lo; ci < c_hi; ci++) {
            const SchurChunk& ch = chunks[ci];
            const uint32_t* local = chunk_local.data() + ci * (size_t)K;
            const uint32_t KL = ch.n_local;
            // window slot of every observation: a point's observations are camera-ascending and so is `local` - one
            // merge walk per point instead of a binary search per observation
            for (uint32_t pp = chunk_p_range[2 * ci]; pp < chunk_p_range[2 * ci + 1]; pp++) {
                size_t sl = 0;
                for (uint32_t q = pt_start[pp]; q < pt_start[pp + 1]; q++) {
                    const int32_t f = h->h_cam_free[q_cam[q]];
                    if (f < 0) continue;
                    while (local[sl] < (uint32_t)f) sl++;
                    q_slot[q] = (uint8_t)sl;
                }
            }
            for (uint32_t b = ch.batch_begin; b < ch.batch_end; b++) {
                SchurBatch& bt = batches[b];
                if (K <= 10) {
                    // the ten-camera kernel's table: the compact row of every (point, window slot) pair of the batch
                    for (uint32_t pp = bt.p_begin; pp < bt.p_end; pp++)
                        for (uint32_t q = pt_start[pp]; q < pt_start[pp + 1]; q++)
                            if (q_slot[q] != 255) pair_row[(size_t)b * kS10PairsPerBatch + (pp - bt.p_begin) * KL + q_slot[q]] = pt_obs[q];
                }
                bool full = bt.p_end - bt.p_begin == (uint32_t)schur_batch_points(K) && bt.q_end - bt.q_begin == (bt.p_end - bt.p_begin) * KL;
                for (uint32_t q = bt.q_begin; full && q < bt.q_end; q++) full = q_slot[q] != 255;
                bt.full = full ? 1u : 0u;
            }
        }
    });
    SETUP_MARK("window slots");
    // long-track points: one slab slot per camera pair and per camera, through the same contribution lists
    std::vector<LongPoint> long_pts;
    std::vector<uint32_t> lo_row, lo_cam, lo_cam_off, pair_a, pair_b, pair_off;
    {
        std::vector<uint32_t> lf;   // free camera of each long observation of the current point
        for (uint32_t p = n_short; p < n_pt; p++) {
            LongPoint lp{p, (uint32_t)lo_row.size(), 0, 0};
            lf.clear();
            for (uint32_t q = pt_start[p]; q < pt_start[p + 1]; q++) {
                const int32_t f = h->h_cam_free[v_obs_cam[pt_obs[q]]];
                if (f < 0) continue;
                if (!lf.empty() && lf.back() == (uint32_t)f) {
                    set_last_error("point %u is observed twice by camera %u", h->pt_int2user[p], v_obs_cam[pt_obs[q]]);
                    return SOSLAM_ERR_INVALID_ARGUMENT;
                }
                lf.push_back((uint32_t)f);
                lo_row.push_back(pt_obs[q]);
                lo_cam.push_back(v_obs_cam[pt_obs[q]]);
            }
            lp.lo_end = (uint32_t)lo_row.size();
            const size_t k = lf.size();
            if (slab_count + (uint64_t)k * (k + 1) / 2 * 36 + (uint64_t)k * 6 > 0xFFFFFFF0ull) {
                set_last_error("Schur slab exceeds 32-bit offsets");
                return SOSLAM_ERR_INVALID_ARGUMENT;
            }
            for (size_t a = 0; a < k; a++)
                for (size_t b = a; b < k; b++) {
                    const int32_t blk = find_block(lf[a], lf[b]);
                    pair_a.push_back(lp.lo_begin + (uint32_t)a);
                    pair_b.push_back(lp.lo_begin + (uint32_t)b);
                    pair_off.push_back((uint32_t)slab_count);
                    if (blk >= 0) blk_contrib[(size_t)blk].push_back((uint32_t)slab_count);
                    slab_count += 36;
                }
            for (size_t a = 0; a < k; a++) {
                lo_cam_off.push_back((uint32_t)slab_count);
                cam_contrib[lf[a]].push_back((uint32_t)slab_count);
                slab_count += 6;
            }
            long_pts.push_back(lp);
        }
    }
    h->n_long = (uint32_t)long_pts.size();
    h->n_long_pairs = (uint32_t)pair_a.size();
    h->n_short = n_short;
    if (slab_count > 0xFFFFFFF0ull) { set_last_error("Schur slab exceeds 32-bit offsets"); return SOSLAM_ERR_INVALID_ARGUMENT; }
    std::vector<uint32_t> bc_ptr(h->n_blocks + 1, 0), bc_off, cc_ptr(nf + 1, 0), cc_off;
    for (uint32_t b = 0; b < h->n_blocks; b++) {
        bc_ptr[b + 1] = bc_ptr[b] + (uint32_t)blk_contrib[b].size();
        bc_off.insert(bc_off.end(), blk_contrib[b].begin(), blk_contrib[b].end());
    }
    for (uint32_t f = 0; f < nf; f++) {
        cc_ptr[f + 1] = cc_ptr[f] + (uint32_t)cam_contrib[f].size();
        cc_off.insert(cc_off.end(), cam_contrib[f].begin(), cam_contrib[f].end());
    }
    h->n_chunks = (uint32_t)chunks.size();
    h->n_batches = (uint32_t)batches.size();
    h->n_point_blocks = backsub_blocks(n_pt);   // ba_backsub writes one partial record per workgroup

    SETUP_MARK("contribution lists");
    // solver choice
    h->bw = 0;
    for (uint32_t b = 0; b < h->n_blocks; b++) h->bw = std::max<int>(h->bw, (int)(h->h_blk_col[b] - h->h_blk_row[b]));
    const bool band_fits = h->bw <= kBandMax;
    // Loop closures - the global BA after a pose-graph run, /root/reference/src/pose_graph_optimizer.cpp:95, slam.cpp:156 - put a
    // few blocks far off a band that holds everything else.  The band the PRECONDITIONER factors is then the narrowest one
    // that holds 99 % of the blocks (h->bw); the blocks outside it stay in S, i.e. in the PCG's matrix-vector product, and
    // only leave the factor: k closure blocks perturb the preconditioned matrix by a term of rank <= 12 k, which costs
    // conjugate gradients about that many extra rounds instead of the thousands block-Jacobi needs on a long chain.
    h->off_band = false;
    if (!band_fits && h->n_blocks > 0) {
        std::vector<uint32_t> hist((size_t)h->bw + 1, 0);
        for (uint32_t b = 0; b < h->n_blocks; b++) hist[h->h_blk_col[b] - h->h_blk_row[b]]++;
        uint64_t acc = 0;
        int w = 0;
        for (; w <= h->bw; w++) {
            acc += hist[(size_t)w];
            if (acc * 100 >= (uint64_t)h->n_blocks * 99) break;
        }
        // Only then: a band wider than the cyclic reduction takes (tracks longer than ten cameras on a long chain) cut at that
        // limit leaves out a large share of every point's coupling; the truncated band then loses positive definiteness once the
        // damping is small, and the diagonal compensation that restores it (take_step) stiffens the smooth modes - measured
        // (scripts/offband_probe.py): 120 cameras with 10 % of tracks of length 20: 480 rounds against block-Jacobi's 340 much
        // cheaper iterations; 150 cameras, tracks of 24: 340 against 180.  Such problems keep the block-Jacobi PCG.
        if (w > kCrBandMax) w = std::getenv("SOSLAM_FORCE_OFFBAND") ? kCrBandMax : 0;   // (the switch: development, scripts/wideband_parity_probe.py)
        if (w >= 1 && std::getenv("SOSLAM_NO_CR") == nullptr && std::getenv("SOSLAM_NO_OFFBAND") == nullptr &&
            h->opt.linear_solver != SOSLAM_SOLVER_BAND_CHOLESKY &&
            h->opt.linear_solver != SOSLAM_SOLVER_DENSE_CHOLESKY && !(h->opt.linear_solver == SOSLAM_SOLVER_AUTO && nf * 6 <= (uint32_t)kDenseAutoLimit)) {
            h->off_band = true;
            h->bw_full = h->bw;
            h->bw = w;
        }
    }
    h->solver = h->opt.linear_solver;
    if (h->solver == SOSLAM_SOLVER_AUTO)
        h->solver = band_fits ? SOSLAM_SOLVER_BAND_CHOLESKY
                              : ((nf * 6 <= (uint32_t)kDenseAutoLimit) ? SOSLAM_SOLVER_DENSE_CHOLESKY : SOSLAM_SOLVER_PCG);
    if (h->solver == SOSLAM_SOLVER_BAND_CHOLESKY && !band_fits) {
        set_last_error("band Cholesky requested but the reduced matrix spans %d block diagonals (limit %d)", h->bw, kBandMax);
        return SOSLAM_ERR_INVALID_ARGUMENT;
    }
    h->pcg_band = h->solver == SOSLAM_SOLVER_PCG && (band_fits || h->off_band);

    SETUP_MARK("solver choice");
    // uploads
    UploadBatch up;
    SOSLAM_CHECK(up.add(h->uv, uv));
    SOSLAM_CHECK(up.add(h->obs_pt, v_obs_pt));
    SOSLAM_CHECK(up.add(h->q_cam, q_cam));
    SOSLAM_CHECK(up.add(h->pt_obs, pt_obs));
    SOSLAM_CHECK(up.add(h->tiles, tiles));
    SOSLAM_CHECK(up.add(h->cam_tile_start, cam_tile_start));
    SOSLAM_CHECK(up.add(h->cam_free, h->h_cam_free));
    SOSLAM_CHECK(up.add(h->pt_start, pt_start));
    SOSLAM_CHECK(up.add(h->q_pt, q_pt));
    SOSLAM_CHECK(up.add(h->q_slot, q_slot));
    SOSLAM_CHECK(up.add(h->chunks, chunks));
    SOSLAM_CHECK(up.add(h->batches, batches));
    SOSLAM_CHECK(up.add(h->pair_row, pair_row));
    SOSLAM_CHECK(up.add(h->chunk_slab, chunk_slab));
    SOSLAM_CHECK(up.add(h->chunk_cam, chunk_cam));
    SOSLAM_CHECK(up.add(h->free_cam, free_cam));
    SOSLAM_CHECK(up.add(h->long_pts, long_pts));
    SOSLAM_CHECK(up.add(h->lo_row, lo_row));
    SOSLAM_CHECK(up.add(h->lo_cam, lo_cam));
    SOSLAM_CHECK(up.add(h->lo_cam_off, lo_cam_off));
    SOSLAM_CHECK(up.add(h->pair_a, pair_a));
    SOSLAM_CHECK(up.add(h->pair_b, pair_b));
    SOSLAM_CHECK(up.add(h->pair_off, pair_off));
    SOSLAM_CHECK(h->long_wy.alloc(lo_row.size() * 36));
    SOSLAM_CHECK(up.add(h->blk_contrib_ptr, bc_ptr));
    SOSLAM_CHECK(up.add(h->blk_contrib_off, bc_off));
    SOSLAM_CHECK(up.add(h->cam_contrib_ptr, cc_ptr));
    SOSLAM_CHECK(up.add(h->cam_contrib_off, cc_off));
    SOSLAM_CHECK(h->slab.alloc((size_t)slab_count));
    SOSLAM_CHECK(up.add(h->diag_block, diag_block));
    SOSLAM_CHECK(up.add(h->row_ptr, row_ptr));
    SOSLAM_CHECK(up.add(h->ent_col, ent_col));
    SOSLAM_CHECK(up.add(h->ent_blk, ent_blk));
    SOSLAM_CHECK(up.add(h->ent_trans, ent_trans));
    SOSLAM_CHECK(up.add(h->blk_row, h->h_blk_row));
    SOSLAM_CHECK(up.add(h->blk_col, h->h_blk_col));

    SOSLAM_CHECK(up.flush(h, s, 0));
    SETUP_MARK("uploads");
    // work buffers
    for (int i = 0; i < 2; i++) {
        SOSLAM_CHECK(h->cams[i].alloc((size_t)n_cam * 6));
        SOSLAM_CHECK(h->pts[i].alloc((size_t)n_pt * 3));
    }
    SOSLAM_CHECK(h->campre.alloc((size_t)n_cam * kPoseStride));
    if ((uint64_t)n_obs * kArRow * 8 > 0xFFFFFFF0ull) {
        set_last_error("%u observations: the compact Jacobian array exceeds the 32-bit row offsets of ba_schur", n_obs);
        return SOSLAM_ERR_INVALID_ARGUMENT;
    }
    SOSLAM_CHECK(h->campre_c.alloc((size_t)n_cam * kPoseStride));
    SOSLAM_CHECK(h->cam_part.alloc(5 * (size_t)cam_update_blocks(n_cam)));
    SOSLAM_CHECK(h->ar.alloc((size_t)n_obs * kArRow));
    SOSLAM_CHECK(h->dcw.alloc((size_t)n_cam * 6));
    SOSLAM_CHECK(h->tile_part.alloc((size_t)h->n_tiles * kTileVals));
    SOSLAM_CHECK(h->cost_part.alloc(std::max<size_t>(h->n_tiles, backsub_blocks(n_pt))));   // (one entry per ba_cost tile, or per workgroup of ba_apply_small)
    SOSLAM_CHECK(h->C.alloc((size_t)n_pt * 6));
    SOSLAM_CHECK(h->gp.alloc((size_t)n_pt * 3));
    SOSLAM_CHECK(h->sp.alloc((size_t)n_pt * 3));
    SOSLAM_CHECK(h->Cinv.alloc((size_t)n_pt * 6));
    SOSLAM_CHECK(h->ptfac.alloc((size_t)n_pt * 12));   // kPtFac doubles per point
    SOSLAM_CHECK(h->sc.alloc((size_t)nf * 6));
    SOSLAM_CHECK(h->lc.alloc((size_t)nf * 6));
    SOSLAM_CHECK(h->dc_free.alloc((size_t)nf * 6));
    SOSLAM_CHECK(h->dc_full.alloc((size_t)n_cam * 6));
    SOSLAM_CHECK(h->dp.alloc((size_t)n_pt * 3));
    SOSLAM_CHECK(h->part.alloc((size_t)h->n_point_blocks * 5));
    SOSLAM_CHECK(h->lin_resid.alloc((size_t)nf * 6));
    SOSLAM_CHECK(h->lin_work.alloc(std::max({pcg_work_count(nf), pcg_band_work_count(nf), pcg_multi_work_count(nf)})));
    UploadBatch zeros;   // fills, as one more kernel
    SOSLAM_CHECK(zeros.add_zero(h->dc_free)); SOSLAM_CHECK(zeros.add_zero(h->Cinv));
    SOSLAM_CHECK(zeros.add_zero(h->lin_resid));
    if (h->solver == SOSLAM_SOLVER_DENSE_CHOLESKY) {
        const size_t n6 = (size_t)nf * 6;
        SOSLAM_CHECK(h->dense.alloc(std::max(n6 * n6 + (size_t)div_up(n6, 32) * 32 * 32 + 1024, dense_inverse_fits(nf) ? dense_inverse_count(nf) : (size_t)0)));
    } else {
        h->dense.release();
    }
    h->cr_rounds = h->off_band ? 8 : 1;
    h->comp_share = 0.0;
    h->cr_factor_valid = false;
    h->use_cr = (h->solver == SOSLAM_SOLVER_BAND_CHOLESKY || h->pcg_band) && h->bw >= 1 && h->bw <= kCrBandMax && nf > 0 &&
                std::getenv("SOSLAM_NO_CR") == nullptr;
    if (h->use_cr) {
        SOSLAM_CHECK(h->cr_ws.alloc(cr_count(nf, h->bw)));
        SOSLAM_CHECK(h->cr_ws2.alloc(cr_count(nf, h->bw)));
        SOSLAM_CHECK(h->lag_status.alloc(SC_COUNT + 8));
        SOSLAM_CHECK(h->lag_status.zero(s));
        SOSLAM_CHECK(h->lag_init());   // the second stream and its events: set-up time, not solve time
        h->fac_idx = 0; h->fac_pending = false; h->lag_ok = true; h->lag_rounds = 3; h->last_rel_decrease = 1.0;
        std::vector<int32_t> map(cr_map_count(nf, h->bw));
        cr_build_map(nf, h->bw, h->n_blocks, h->h_blk_row.data(), h->h_blk_col.data(), map.data());
        SOSLAM_CHECK(h->cr_map.upload(map, s));
        if (h->off_band) {
            std::vector<uint32_t> cptr, cent;
            cr_build_comp_lists(nf, h->bw, h->n_blocks, h->h_blk_row.data(), h->h_blk_col.data(), cptr, cent);
            SOSLAM_CHECK(h->cr_comp_ptr.upload(cptr, s));
            SOSLAM_CHECK(h->cr_comp_ent.upload(cent, s));
            SOSLAM_CHECK(h->cr_comp.alloc((size_t)nf * 6));
        }
        h->band.release(); h->bandT.release(); h->band_dinv.release();
    } else if (h->solver == SOSLAM_SOLVER_BAND_CHOLESKY || h->pcg_band) {
        h->cr_ws.release();
        SOSLAM_CHECK(h->band.alloc(band_count(nf, h->bw)));
        SOSLAM_CHECK(h->bandT.alloc(band_count(nf, h->bw)));
        SOSLAM_CHECK(h->bandT.zero(s));
        SOSLAM_CHECK(h->band_dinv.alloc((size_t)nf * 36));
    } else {
        h->band.release();
        h->bandT.release();
        h->band_dinv.release();
    }
    // Two-level PCG where block-Jacobi PCG would run: the band is wider than either band factor takes, or more than 1 % of the
    // blocks lie outside it.  The weak directions of S on a camera chain are its drift modes - rigid motions of whole stretches
    // of cameras - which block-Jacobi sees one camera at a time (400 - 500 iterations on configs[2] with 10 % of tracks of length
    // 20); six rigid-body modes per aggregate of consecutive free cameras span them.
    h->two_level = h->solver == SOSLAM_SOLVER_PCG && !h->pcg_band && nf >= kPcgMultiMinRows && std::getenv("SOSLAM_NO_TWO_LEVEL") == nullptr;
    if (h->two_level) {
        // cameras per aggregate: at most 200 aggregates (1 200 coarse unknowns).  Measured on configs[2] with 10 % of tracks of length 20
        // (scripts/tl_agg_probe.py): aggregates of 12 / 6 / 4 cameras take 68 / 53 / 46 PCG iterations per solve, 2.84 / 2.50 / 2.46 ms per
        // LM iteration - the coarse inverse grows as fast as the iterations shrink.  (Letting that inverse lag one solve behind, as
        // the pose graph does, is NOT an option here: the damping on S changes threefold from one iteration to the next, and a
        // coarse operator of the previous matrix sent the PCG to 400 - 1 300 iterations.)
        uint32_t per = std::max<uint32_t>(6, (nf + 199) / 200);
        if (const char* e = std::getenv("SOSLAM_TL_AGG")) per = std::max<uint32_t>((uint32_t)std::max(1, std::atoi(e)), (nf + 199) / 200);   // development
        if (per > 42) h->two_level = false;                              // one workgroup of pcg2 per aggregate: 42 block rows
        else {
            std::vector<uint32_t> agg_ptr{0}, row_agg(nf), agg_ref;
            for (uint32_t f0 = 0; f0 < nf; f0 += per) {
                const uint32_t f1 = std::min(nf, f0 + per);
                for (uint32_t f = f0; f < f1; f++) row_agg[f] = (uint32_t)agg_ref.size();
                agg_ref.push_back(free_cam[f0]);
                agg_ptr.push_back(f1);
            }
            h->tl_n_agg = (uint32_t)agg_ref.size();
            h->tl_ncp = (h->tl_n_agg * 6 + 59) / 60 * 60;
            std::vector<uint32_t> cb_ptr, cb_ent, cb_I, cb_J;
            two_level_lists(h->tl_n_agg, row_agg.data(), h->n_blocks, h->h_blk_row.data(), h->h_blk_col.data(), cb_ptr, cb_ent, cb_I, cb_J);
            h->tl_n_cb = (uint32_t)cb_I.size();
            SOSLAM_CHECK(h->tl_agg_ptr.upload(agg_ptr, s));
            SOSLAM_CHECK(h->tl_row_agg.upload(row_agg, s));
            SOSLAM_CHECK(h->tl_agg_ref.upload(agg_ref, s));
            SOSLAM_CHECK(h->tl_cb_ptr.upload(cb_ptr, s));
            SOSLAM_CHECK(h->tl_cb_ent.upload(cb_ent, s));
            SOSLAM_CHECK(h->tl_cb_I.upload(cb_I, s));
            SOSLAM_CHECK(h->tl_cb_J.upload(cb_J, s));
            SOSLAM_CHECK(h->tl_P.alloc((size_t)nf * 36));
            SOSLAM_CHECK(h->tl_G.alloc((size_t)h->tl_n_agg * 36));
            SOSLAM_CHECK(h->tl_Ac0.alloc((size_t)h->tl_ncp * h->tl_ncp));
            SOSLAM_CHECK(h->tl_Ainv.alloc((size_t)h->tl_ncp * h->tl_ncp));
            SOSLAM_CHECK(h->tl_ebuf.alloc(2 * 3600));
            SOSLAM_CHECK(h->tl_rc.alloc(h->tl_ncp));
            SOSLAM_CHECK(h->tl_rc.zero(s));
            SOSLAM_CHECK(h->tl_status.alloc(8));
            SOSLAM_CHECK(h->tl_status.zero(s));
            SOSLAM_CHECK(h->lin_work.alloc(std::max({pcg_work_count(nf), pcg_band_work_count(nf), pcg_multi_work_count(nf), pcg2_work_count(nf, h->tl_n_agg)})));
            h->tl_last_it = 0;
        }
    }
    h->reduce_main = (uint64_t)h->n_blocks * 36 + (uint64_t)nf * 18 + 4;
    h->reduce_count = h->reduce_main + SC_COUNT;
    SOSLAM_CHECK(h->reduce_own.alloc(h->reduce_count));
    SOSLAM_CHECK(zeros.add_zero(h->reduce_own));
    SOSLAM_CHECK(zeros.flush(h, s, 1));
    h->reduce = h->reduce_own.p;
    if (!h->host_raw) {
        SOSLAM_CHECK(h->pinned_alloc(reinterpret_cast<void**>(&h->host_raw), sizeof(double) * (SC_COUNT + 4 + 1)));
        h->host_scal = h->host_raw + 4;
        h->host_seq = reinterpret_cast<unsigned long long*>(h->host_raw + 4 + SC_COUNT);
        *h->host_seq = 0;
    }
    SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
    h->have_problem = true;
    h->have_state = false;
    SETUP_MARK("alloc+sync");
    h->setup_seconds = now_sec() - t0;
    return SOSLAM_OK;
}

// ---- device pipeline ---------------------------------------------------------------------------------

// Structure-only problems small enough for one workgroup take a whole LM iteration in one launch (ba_points_step): no
// separate linearisation, no Schur complement, no camera system.  Not in multi-rank jobs (their sums cross ranks).
bool points_only(const soslam_ba* h) { return h->n_free == 0 && h->n_pt > 0 && h->n_pt <= kPointsOnlyMax && !h->collective(); }

// residuals, Jacobians and the J^T J / J^T r blocks at the current state.  in_lm_loop: called by the LM loop, which on
// the structure-only path needs no linearisation of its own (every ba_points_step launch linearises at its x)
// The ten-camera Schur kernel can form the point blocks C = sum J_p^T J_p, g_p = sum J_p^T r itself (every point goes
// through it: no long tracks, windows of at most ten cameras): ba_point_reduce then runs only where something else
// needs its output first - the Jacobi scales of the first linearisation.
bool points_fused(const soslam_ba* h)
{
    static const bool off = std::getenv("SOSLAM_NO_POINT_FUSE") != nullptr;   // development / tests: always the separate point pass
    return !off && h->kmax <= 10 && h->n_long == 0 && h->n_chunks > 0 && !points_only(h);
}

int linearize(soslam_ba* h, bool in_lm_loop = false)
{
    hipStream_t s = h->stream;
    if (in_lm_loop && points_only(h)) { h->linearized = true; h->campre_current = false; return SOSLAM_OK; }
    {
        StageScope sc(h, SOSLAM_STAGE_LINEARIZE);
        // after an accepted step the candidate's pose table is already in place (swapped in by the LM loop)
        if (!h->campre_current) launch_pose_prepare(s, h->n_cam, h->cams[h->cur].p, h->campre.p);
        h->campre_current = false;
        launch_linearize(s, h->n_tiles, h->tiles.p, h->uv.p, h->obs_pt.p, h->campre.p, h->pts[h->cur].p, h->cam_free.p,
                         h->proj, h->opt.huber_delta, h->rows(), h->tile_part.p);
        // the cost at this point: known on the host after an accepted step (it was the candidate's cost, summed over
        // ranks); summed from the tiles only for a state the loop has not evaluated yet
        if (!h->x_cost_known) launch_sum_strided(s, h->tile_part.p, h->n_tiles, kTileVals, 27, 0.5, h->scalp() + SC_COST_X);
    }
    if (!points_fused(h) || !h->scale_init) {
        StageScope sc(h, SOSLAM_STAGE_POINT_REDUCE);
        launch_point_reduce(s, h->n_pt, h->pt_start.p, h->pt_obs.p, h->q_cam.p, h->rows(), h->campre.p, h->C.p, h->gp.p);
        if (!h->scale_init) launch_point_scale(s, h->n_pt, h->C.p, h->opt.jacobi_scaling, h->sp.p);
        h->pt_blocks_valid = true;
    } else {
        h->pt_blocks_valid = false;   // the ten-camera Schur kernel forms C and gp from the rows (see launch_schur)
    }
    SOSLAM_HIP_CHECK(hipGetLastError());
    h->linearized = true;
    return SOSLAM_OK;
}

// spin on a sequence number in pinned host memory that a kernel on the handle's stream writes
int wait_host_seq(soslam_ba* h, const unsigned long long* word, unsigned long long seq)
{
    SOSLAM_HIP_CHECK(hipGetLastError());
    uint64_t spins = 0;
    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xFFFF) == 0) {   // a failed launch or a dead device must not spin forever
            const hipError_t q = hipStreamQuery(h->stream);
            if (q != hipErrorNotReady && q != hipSuccess) SOSLAM_HIP_CHECK(q);
            if (q == hipSuccess && __atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) {
                set_last_error("a sequence number published by the device did not arrive");
                return SOSLAM_ERR_HIP;
            }
        }
    }
    return SOSLAM_OK;
}

// pinned, host-coherent staging memory of the host-collective leg (+ its sequence word)
int stage_reserve(soslam_ba* h, uint64_t count)
{
    if (h->stage_count >= count) return SOSLAM_OK;
    SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->stage) (void)hipHostFree(h->stage);
    h->stage = nullptr; h->stage_count = 0;
    // one allocation: [count f64 | sequence word]; coherent (uncached on the device): the device reads what the host wrote
    SOSLAM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->stage), sizeof(double) * (count + 1), hipHostMallocCoherent | hipHostMallocMapped));
    h->stage_count = count;
    h->stage_seq = reinterpret_cast<unsigned long long*>(h->stage + count);
    *h->stage_seq = 0;
    h->stage_seq_next = 0;
    return SOSLAM_OK;
}

int do_allreduce(soslam_ba* h, double* buf, uint64_t count, int op)
{
    // the library's own RCCL leg: in place, on the handle's stream - ordered against its kernels by construction
    if (h->rccl) return rccl_allreduce_f64(h->rccl, buf, count, op, h->stream);
    if (h->host_allreduce) {
        // host collective (MPI without device support, gloo): the library stages the range through its own pinned buffer on
        // its own stream - device -> host, the caller's in-place reduction of the host range, host -> device.  Both
        // directions are kernels on the handle's stream (pinned memory is mapped into the device) and the host learns of
        // the download by polling a sequence number a kernel writes behind it - the mechanism of the step-scalar
        // publication: no copy engine, no second queue, no stream-synchronisation call inside an iteration.
        SOSLAM_CHECK(stage_reserve(h, count));
        double* stage_dev = nullptr;
        unsigned long long* seq_dev = nullptr;
        SOSLAM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&stage_dev), h->stage, 0));
        SOSLAM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&seq_dev), h->stage_seq, 0));
        launch_copy_f64(h->stream, stage_dev, buf, count);
        const unsigned long long seq = ++h->stage_seq_next;
        launch_flag(h->stream, seq_dev, seq);
        SOSLAM_CHECK(wait_host_seq(h, h->stage_seq, seq));
        if (h->host_allreduce(h->host_allreduce_user, h->stage, count, op) != 0) {
            set_last_error("host all-reduce callback failed (rank %d, %llu f64)", h->rank, (unsigned long long)count);
            return SOSLAM_ERR_COMM;
        }
        // the upload is ordered before every later kernel by the stream, and before the next download into the staging
        // buffer likewise: the host never has to wait for it
        launch_copy_f64(h->stream, buf, stage_dev, count);
        return SOSLAM_OK;
    }
    if (!h->allreduce) return SOSLAM_OK;
    if (h->allreduce(h->allreduce_user, buf, count, op, h->stream) != 0) {
        set_last_error("all-reduce callback failed (rank %d, %llu f64)", h->rank, (unsigned long long)count);
        return SOSLAM_ERR_COMM;
    }
    return SOSLAM_OK;
}

BsrView bsr_view(const soslam_ba* h)
{
    return BsrView{h->n_free, h->row_ptr.p, h->ent_col.p, h->ent_blk.p, h->ent_trans.p, h->diag_block.p, h->S()};
}

// point elimination: windowed chunks, then the long-track points; both write slab slots for ba_schur_reduce
void run_schur(soslam_ba* h, const LmDiag& lm)
{
    hipStream_t s = h->stream;
    launch_schur(s, h->kmax, h->n_chunks, h->chunks.p, h->batches.p, h->chunk_slab.p, h->chunk_cam.p, h->pair_row.p, h->pt_obs.p, h->q_pt.p,
                 h->q_slot.p, h->rows(), h->campre.p, h->pts[h->cur].p, h->C.p, h->gp.p, h->sp.p, lm, h->Cinv.p, h->ptfac.p, h->slab.p, h->scalp(),
                 h->pt_start.p, h->q_cam.p, h->pt_blocks_valid ? 0 : 1);
    h->pt_blocks_valid = true;
    launch_schur_long(s, h->n_long, h->long_pts.p, h->lo_row.p, h->lo_cam.p, h->lo_cam_off.p, h->n_long_pairs, h->pair_a.p, h->pair_b.p,
                      h->pair_off.p, h->ar.p, h->campre.p, h->pts[h->cur].p, h->C.p, h->gp.p, h->sp.p, lm, h->Cinv.p, h->long_wy.p,
                      h->slab.p, h->scalp());
}


// one trust-region step from the current linearisation: reduced system, solve, candidate, candidate cost
int take_step(soslam_ba* h, double radius, bool speculate = false, bool stop_vote = false)
{
    hipStream_t s = h->stream;
    unsigned long long published = 0;   // sequence number if the scalars were already handed to the host
    bool speculated = false;            // the linearisation at the candidate is enqueued behind the acceptance test
    const LmDiag lm = lm_diag(h, radius);
    if (points_only(h)) {
        // every camera constant: one launch per LM iteration (see PointsStepArgs)
        if (!h->points_only_ready) {
            launch_pose_prepare(s, h->n_cam, h->cams[h->cur].p, h->campre.p);
            launch_pose_prepare(s, h->n_cam, h->cams[h->cur].p, h->campre_c.p);
            SOSLAM_HIP_CHECK(hipMemcpyAsync(h->cams[h->cur ^ 1].p, h->cams[h->cur].p, sizeof(double) * 6 * h->n_cam, hipMemcpyDeviceToDevice, s));
            SOSLAM_CHECK(h->dc_full.zero(s));
            h->points_only_ready = true;
        }
        StageScope sc(h, SOSLAM_STAGE_BACKSUB);
        PointsStepArgs a{};
        a.n_pt = h->n_pt; a.pt_start = h->pt_start.p; a.pt_obs = h->pt_obs.p; a.q_cam = h->q_cam.p; a.uv = h->uv.p;
        a.campre = h->campre.p; a.pts = h->pts[h->cur].p; a.pts_out = h->pts[h->cur ^ 1].p; a.dp = h->dp.p;
        a.C = h->C.p; a.gp = h->gp.p; a.sp = h->sp.p; a.Cinv = h->Cinv.p; a.lm = lm; a.huber_delta = h->opt.huber_delta;
        a.bound_lo = h->opt.lower_bound; a.bound_hi = h->opt.upper_bound; a.init_scale = h->scale_init ? 0 : 1;
        a.jacobi = h->opt.jacobi_scaling; a.scal = h->scalp(); a.cost_x_out = h->tail(); a.x_cost = h->x_cost;
        a.min_relative_decrease = h->opt.min_relative_decrease; a.gate_enabled = 0;
        (void)speculate; (void)stop_vote;
        h->scale_init = true;
        published = ++h->publish_seq;
        launch_points_step(s, a, h->proj, h->tail(), 4 + SC_COUNT, 4 + SC_LIN_ITERS, 4, h->host_raw, h->host_seq, published);
        SOSLAM_HIP_CHECK(hipGetLastError());
        return wait_host_seq(h, h->host_seq, published);
    }
    bool damp_fused = false, apply_fused = false, sums_fused = false;
    {
        StageScope sc(h, SOSLAM_STAGE_SCHUR);
        run_schur(h, lm);
        if (h->fac_pending) {
            // the refresh of the lagged factor (second stream) reads S and writes the other workspace: S is rewritten below,
            // and from here on that workspace is the newest factor
            SOSLAM_HIP_CHECK(hipStreamWaitEvent(s, h->ev_factor, 0));
            h->fac_pending = false;
            h->fac_idx ^= 1;
        }
        // one rank and a dense solve (the reference's sliding windows): the camera damping rides along - the diagonal of B is
        // this rank's, complete, and nothing between here and the solve needs S undamped
        damp_fused = !h->collective() && h->n_free && h->solver == SOSLAM_SOLVER_DENSE_CHOLESKY;
        const CamDamp fd{h->diagB(), h->sc.p, h->lc.p, h->diag_block.p, lm, h->scale_init ? 0 : 1, h->opt.jacobi_scaling, h->n_free};
        launch_schur_reduce(s, h->n_blocks, h->n_free, h->blk_contrib_ptr.p, h->blk_contrib_off.p, h->cam_contrib_ptr.p,
                            h->cam_contrib_off.p, h->blk_row.p, h->blk_col.p, h->free_cam.p, h->campre.p, h->slab.p, h->cam_tile_start.p, h->tile_part.p,
                            h->S(), h->rhs(), h->diagB(), h->gc_red(), h->scalp() + SC_COST_X, h->tail(), damp_fused ? &fd : nullptr);
    }
    {
        StageScope sc(h, SOSLAM_STAGE_ALLREDUCE);
        SOSLAM_CHECK(do_allreduce(h, h->reduce, h->reduce_main, SOSLAM_REDUCE_SUM));
    }
    {
        StageScope sc(h, SOSLAM_STAGE_SOLVE);
        // camera damping onto S: a kernel of its own, except on the cyclic-reduction path, whose gather applies it
        const bool damp_in_gather = h->use_cr && h->n_free && (h->solver == SOSLAM_SOLVER_BAND_CHOLESKY || h->pcg_band) &&
                                    h->solver != SOSLAM_SOLVER_DENSE_CHOLESKY;
        const CamDamp damp{h->diagB(), h->sc.p, h->lc.p, h->diag_block.p, lm, h->scale_init ? 0 : 1, h->opt.jacobi_scaling, h->n_free};
        if (!damp_in_gather && !damp_fused)
            launch_cam_damp(s, h->n_free, h->diagB(), h->sc.p, h->scale_init ? 0 : 1, h->opt.jacobi_scaling, lm, h->diag_block.p,
                            h->S(), h->lc.p);
        h->scale_init = true;
        const double* resid = nullptr;
        if (h->n_free) {
            if (h->solver == SOSLAM_SOLVER_DENSE_CHOLESKY) {
                static const bool no_dense2 = std::getenv("SOSLAM_NO_DENSE2") != nullptr;   // development / tests: the blocked Cholesky
                if (dense2_fits(h->n_free) && !no_dense2) {
                    launch_dense2_solve(s, bsr_view(h), h->n_blocks, h->blk_row.p, h->blk_col.p, h->rhs(), h->dc_free.p, h->scalp());
                } else if (dense_small_fits(h->n_free)) {
                    launch_dense_small_solve(s, bsr_view(h), h->n_blocks, h->blk_row.p, h->blk_col.p, h->rhs(), h->dc_free.p, h->scalp());
                } else if (dense_inverse_fits(h->n_free) && std::getenv("SOSLAM_DENSE_CHOLESKY") == nullptr) {
                    launch_dense_inverse_solve(s, bsr_view(h), h->n_blocks, h->blk_row.p, h->blk_col.p, h->dense.p, h->rhs(), h->dc_free.p, h->scalp());
                } else {
                    launch_bsr_to_dense(s, bsr_view(h), h->n_blocks, h->blk_row.p, h->blk_col.p, h->dense.p);
                    launch_dense_cholesky_solve(s, h->n_free * 6, h->dense.p, h->rhs(), h->dc_free.p, h->scalp());
                }
            } else if (h->use_cr && h->solver == SOSLAM_SOLVER_BAND_CHOLESKY) {
                launch_cr_factor(s, bsr_view(h), h->cr_map.p, h->bw, h->cr_ws.p, h->scalp(), &damp, h->rhs());
                launch_cr_solve(s, h->n_free, h->bw, h->cr_ws.p, h->rhs(), h->dc_free.p, nullptr, true);
            } else if (h->use_cr && h->pcg_band) {
                // LAGGED FACTOR.  Once the iterates have all but converged (the last accepted step lowered the cost by less than
                // 1e-7 of it) S hardly moves from one linearisation to the next, and the factor of the PREVIOUS reduced matrix is
                // a preconditioner that takes PCG to the tolerance in two or three rounds (measured, configs[2]: 8 rounds and more
                // during the first twelve iterations, 3 from the thirteenth, 2 from the sixteenth).  A round is 54 us, the
                // factorisation 170: so in that phase this iteration's factorisation leaves the critical path - it runs on a
                // second stream beside the rest of the iteration, for the NEXT solve - and the solve is the PCG alone.  Early
                // iterations, rejected or invalid steps and anything that made the last lagged solve take more than four rounds
                // get a fresh factor as before.
                static const bool lag_off = std::getenv("SOSLAM_NO_LAG") != nullptr;
                const bool use_lag = !lag_off && !h->off_band && h->cr_factor_valid && h->lag_ok && h->last_rel_decrease < 1e-7;
                if (use_lag) {
                    double* ws_old = h->cr_ws_of(h->fac_idx);
                    double* ws_new = h->cr_ws_of(h->fac_idx ^ 1);
                    launch_cam_damp(s, h->n_free, h->diagB(), h->sc.p, damp.init_scale, h->opt.jacobi_scaling, lm, h->diag_block.p, h->S(), h->lc.p);
                    // the refresh, off the critical path: S (damped) is complete here; nothing writes it again before the next
                    // iteration's ba_schur_reduce, which waits for ev_factor (take_step, above)
                    SOSLAM_CHECK(h->lag_init());
                    SOSLAM_HIP_CHECK(hipEventRecord(h->ev_s_ready, s));
                    SOSLAM_HIP_CHECK(hipStreamWaitEvent(h->fstream, h->ev_s_ready, 0));
                    launch_cr_factor(h->fstream, bsr_view(h), h->cr_map.p, h->bw, ws_new, h->lag_status.p, nullptr, nullptr);
                    SOSLAM_HIP_CHECK(hipEventRecord(h->ev_factor, h->fstream));
                    h->fac_pending = true;
                    launch_pcg_cr(s, bsr_view(h), h->bw, ws_old, h->rhs(), h->dc_free.p, h->lin_resid.p, h->lin_work.p, h->opt.pcg_tolerance,
                                  std::min(h->opt.pcg_max_iterations, h->lag_rounds), h->scalp(), false);
                    h->lagged_solve = true;
                } else {
                h->lagged_solve = false;
                if (!h->off_band) {
                    launch_cr_factor(s, bsr_view(h), h->cr_map.p, h->bw, h->cr_ws_of(h->fac_idx), h->scalp(), &damp, h->rhs());
                } else {
                    // The factored band leaves blocks of S out, and the band part of a positive definite matrix need not be
                    // positive definite.  A breakdown must not become an invalid LM step (the reference's direct solver has
                    // none): the host looks at the factorisation's status word - one synchronisation per iteration, in this mode
                    // only - and refactors with a larger share of the left-out blocks' row sums on the diagonal (kept for the
                    // following iterations; at share 1 the factored matrix is positive definite whatever was left out).
                    launch_cr_comp(s, h->S(), h->n_free, h->cr_comp_ptr.p, h->cr_comp_ent.p, h->cr_comp.p);
                    for (int attempt = 0; attempt < 6; attempt++) {
                        launch_cr_factor(s, bsr_view(h), h->cr_map.p, h->bw, h->cr_ws_of(h->fac_idx), h->scalp(), attempt == 0 ? &damp : nullptr,
                                         h->rhs(), h->cr_comp.p, h->comp_share);
                        double st = 0.0;
                        SOSLAM_HIP_CHECK(hipMemcpyAsync(&st, h->scalp() + SC_LIN_STATUS, sizeof st, hipMemcpyDeviceToHost, s));
                        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
                        if (st == 0.0 || h->comp_share >= 1.0) break;
                        SOSLAM_HIP_CHECK(hipMemsetAsync(h->scalp() + SC_LIN_STATUS, 0, sizeof(double), s));
                        h->comp_share = h->comp_share == 0.0 ? 0.125 : std::min(1.0, 2.0 * h->comp_share);
                    }
                }
                launch_pcg_cr(s, bsr_view(h), h->bw, h->cr_ws_of(h->fac_idx), h->rhs(), h->dc_free.p, h->lin_resid.p, h->lin_work.p,
                              h->opt.pcg_tolerance, std::min(h->opt.pcg_max_iterations, h->cr_rounds), h->scalp(), true);
                h->cr_factor_valid = true;
                if (h->off_band) {
                    // this mode synchronises anyway (above): the solve runs to its tolerance here, in chunks of as many rounds as
                    // the last solve used, instead of handing an unconverged step to the LM loop
                    int enq = std::min(h->opt.pcg_max_iterations, h->cr_rounds);
                    while (enq < h->opt.pcg_max_iterations) {
                        double lin[3];
                        SOSLAM_HIP_CHECK(hipMemcpyAsync(lin, h->scalp() + SC_LIN_ITERS, sizeof lin, hipMemcpyDeviceToHost, s));
                        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
                        if (lin[1] <= h->opt.pcg_tolerance || lin[2] != 0.0) break;
                        const int more = std::min(h->opt.pcg_max_iterations - enq, std::max(4, enq / 2));
                        launch_pcg_cr_more(s, bsr_view(h), h->bw, h->cr_ws_of(h->fac_idx), h->dc_free.p, h->lin_resid.p, h->lin_work.p, h->opt.pcg_tolerance, more,
                                           h->scalp());
                        enq += more;
                    }
                }
                }
                resid = h->lin_resid.p;
            } else if (h->solver == SOSLAM_SOLVER_BAND_CHOLESKY) {
                launch_bsr_to_band(s, bsr_view(h), h->n_blocks, h->blk_row.p, h->blk_col.p, h->bw, h->band.p);
                launch_band_cholesky(s, h->n_free, h->bw, h->band.p, h->bandT.p, h->band_dinv.p, h->scalp());
                launch_band_solve(s, h->n_free, h->bw, h->band.p, h->bandT.p, h->band_dinv.p, h->rhs(), h->dc_free.p);
            } else if (h->pcg_band) {
                launch_bsr_to_band(s, bsr_view(h), h->n_blocks, h->blk_row.p, h->blk_col.p, h->bw, h->band.p);
                launch_band_cholesky(s, h->n_free, h->bw, h->band.p, h->bandT.p, h->band_dinv.p, h->scalp());
                launch_pcg_band(s, bsr_view(h), h->bw, h->band.p, h->bandT.p, h->band_dinv.p, h->rhs(), h->dc_free.p,
                                h->lin_resid.p, h->lin_work.p, h->opt.pcg_tolerance, std::min(h->opt.pcg_max_iterations, 4),
                                h->scalp());
                resid = h->lin_resid.p;
            } else if (h->n_free >= kPcgMultiMinRows) {
                // no band to factor: PCG with every vector operation spread over many workgroups - block-Jacobi plus, when set up
                // (build_problem), the rigid-body coarse space
                double rel = 0.0;
                int it;
                if (h->two_level) {
                    launch_ba_coarse_basis(s, h->n_free, h->free_cam.p, h->tl_row_agg.p, h->tl_agg_ref.p, h->campre.p, h->tl_P.p);
                    launch_coarse_assemble(s, h->tl_n_agg, h->tl_agg_ptr.p, h->tl_n_cb, h->tl_cb_ptr.p, h->tl_cb_ent.p, h->tl_cb_I.p, h->tl_cb_J.p,
                                           h->blk_row.p, h->blk_col.p, h->S(), h->tl_P.p, h->tl_ncp, h->tl_G.p, h->tl_Ac0.p);
                    pcg2_coarse_inverse(s, h->tl_n_agg, h->tl_ncp, h->tl_Ac0.p, h->tl_G.p, 0.0, h->tl_Ainv.p, h->tl_ebuf.p, h->tl_status.p);
                    const TwoLevelView tl{h->tl_n_agg, h->tl_ncp, h->tl_agg_ptr.p, h->tl_P.p, h->tl_Ainv.p, h->tl_rc.p};
                    it = pcg2_solve(s, bsr_view(h), 0.0, h->rhs(), h->dc_free.p, h->lin_resid.p, h->lin_work.p, tl, h->opt.pcg_tolerance,
                                    h->opt.pcg_max_iterations, h->tl_last_it > 0 ? h->tl_last_it + 2 : 48, &rel);
                    if (it > 0) h->tl_last_it = it;
                } else
                it = pcg_multi_solve(s, bsr_view(h), 0.0, h->rhs(), h->dc_free.p, h->lin_resid.p, h->lin_work.p,
                                     h->opt.pcg_tolerance, h->opt.pcg_max_iterations, 16, &rel);
                const double lin[3] = {(double)std::max(it, 0), rel, it < 0 ? 2.0 : 0.0};   // SC_LIN_ITERS, _RESID, _STATUS
                static_assert(SC_LIN_RESID == SC_LIN_ITERS + 1 && SC_LIN_STATUS == SC_LIN_ITERS + 2, "scalar slots are contiguous");
                SOSLAM_HIP_CHECK(hipMemcpyAsync(h->scalp() + SC_LIN_ITERS, lin, sizeof lin, hipMemcpyHostToDevice, s));
                SOSLAM_HIP_CHECK(hipStreamSynchronize(s));   // lin[] lives on this stack frame
                resid = h->lin_resid.p;
            } else {
                launch_pcg(s, bsr_view(h), h->rhs(), h->dc_free.p, h->lin_resid.p, h->lin_work.p, h->opt.pcg_tolerance,
                           h->opt.pcg_max_iterations, h->scalp());
                resid = h->lin_resid.p;
            }
        }
        // few cameras (the reference's windows), one rank: candidate cameras, back-substitution and candidate cost are ONE launch
        static const bool no_fuse = std::getenv("SOSLAM_NO_APPLY_FUSE") != nullptr;   // development / tests
        apply_fused = !no_fuse && !h->collective() && apply_small_fits(h->n_cam, h->n_pt);
        if (apply_fused) {
            // ... and the workgroup that finishes last sums the step scalars, tests acceptance and publishes (ba_step_sums' work)
            static const bool no_sums = std::getenv("SOSLAM_NO_SUMS_FUSE") != nullptr;   // development / tests
            sums_fused = !no_sums;
            StepSumsLaunch sl{};
            if (sums_fused) {
                if (!h->arrivals.p) { SOSLAM_CHECK(h->arrivals.alloc(4)); SOSLAM_CHECK(h->arrivals.zero(s)); }
                published = ++h->publish_seq;
                const bool spec = speculate && h->x_cost_known;
                sl = StepSumsLaunch{h->scalp() + SC_MCC_PTS, h->scalp() + SC_MCC_CAM, h->scalp() + SC_CAND_COST, h->scalp() + SC_GATE,
                                    h->scalp() + SC_LIN_ITERS, h->x_cost, h->opt.min_relative_decrease, spec ? 1 : 0, stop_vote ? 1.0 : 0.0,
                                    h->tail(), 4 + SC_COUNT, 4 + SC_LIN_ITERS, 4, h->host_raw, h->host_seq, published, is_constrained(h) ? 1 : 0};
                speculated = spec;
            }
            launch_apply_small(s, h->n_cam, h->cam_free.p, h->cams[h->cur].p, h->dc_free.p, h->lc.p, h->gc_red(), resid, h->cams[h->cur ^ 1].p,
                               h->dc_full.p, h->dcw.p, h->cam_part.p, h->campre_c.p, h->n_pt, h->pt_start.p, h->pt_obs.p, h->q_cam.p, h->ar.p,
                               h->campre.p, h->Cinv.p, h->C.p, h->gp.p, h->sp.p, h->pts[h->cur].p, lm, h->opt.lower_bound, h->opt.upper_bound,
                               h->pts[h->cur ^ 1].p, h->dp.p, h->part.p, h->uv.p, h->proj, h->opt.huber_delta, h->cost_part.p, h->arrivals.p,
                               sums_fused ? &sl : nullptr);
        } else
        launch_cam_update(s, h->n_cam, h->cam_free.p, h->cams[h->cur].p, h->dc_free.p, h->lc.p, h->gc_red(), resid, h->campre.p,
                          h->cams[h->cur ^ 1].p, h->dc_full.p, h->dcw.p, h->cam_part.p, h->campre_c.p);
    }
    if (!apply_fused) {
        StageScope sc(h, SOSLAM_STAGE_BACKSUB);
        launch_backsub(s, h->n_pt, h->pt_start.p, h->pt_obs.p, h->q_cam.p, h->ar.p, h->campre.p, h->dcw.p, h->Cinv.p, h->C.p,
                       h->gp.p, h->sp.p, h->pts[h->cur].p, lm, h->opt.lower_bound, h->opt.upper_bound, h->pts[h->cur ^ 1].p,
                       h->dp.p, h->part.p);
    }
    if (!sums_fused) {
        StageScope sc(h, SOSLAM_STAGE_COST);
        // the candidate has its own pose table (written by ba_cam_update): campre stays at the linearisation point, the
        // compact rows need it; an accepted step swaps the two tables instead of preparing the same poses again
        if (!apply_fused)
        launch_cost(s, h->n_tiles, h->tiles.p, h->uv.p, h->obs_pt.p, h->campre_c.p, h->pts[h->cur ^ 1].p, h->proj,
                    h->opt.huber_delta, h->cost_part.p);
        // one launch for the back-substitution's step scalars and the candidate cost; on a single rank nothing follows
        // it, so it also hands the iteration's scalars to the host
        if (!h->collective()) published = ++h->publish_seq;
        // speculation (cost at x known): the acceptance test runs on the device and the linearisation at the candidate is
        // enqueued right behind it, gated by that decision - the GPU does not wait for the host's round trip.  On one rank
        // the sum kernel decides; in a multi-rank job the same test runs after the scalars were summed (ba_gate_publish)
        const bool spec = speculate && h->x_cost_known;
        static_assert(SC_LIN_STATUS == SC_LIN_ITERS + 2 && SC_SCHUR_STATUS == SC_LIN_ITERS + 3, "status words as StepGate reads them");
        static_assert(SC_STOP == SC_MCC_PTS + 5 && SC_STOP == SC_GMAX_PTS + 1, "the stop vote follows the five step scalars");
        launch_step_sums(s, h->part.p, h->n_point_blocks, h->scalp() + SC_MCC_PTS, h->cam_part.p, cam_update_blocks(h->n_cam),
                         h->scalp() + SC_MCC_CAM, h->cost_part.p, apply_fused ? h->n_point_blocks : h->n_tiles, h->scalp() + SC_CAND_COST, h->scalp() + SC_GATE,
                         h->scalp() + SC_LIN_ITERS, h->x_cost, h->opt.min_relative_decrease, (spec && !h->collective()) ? 1 : 0,
                         stop_vote ? 1.0 : 0.0, h->tail(), 4 + SC_COUNT, 4 + SC_LIN_ITERS, 4, published ? h->host_raw : nullptr, h->host_seq,
                         published, is_constrained(h) ? 1 : 0);
        speculated = spec;
    }
    static_assert(4 + SC_COUNT <= 64, "one wave publishes the scalars");
    if (h->collective()) {
        StageScope sc(h, SOSLAM_STAGE_ALLREDUCE);
        // a rank whose point elimination failed turns its share of the candidate cost into +inf: every rank rejects alike
        launch_status_poison(s, h->scalp());
        SOSLAM_CHECK(do_allreduce(h, h->scalp() + SC_CAND_COST, 5, SOSLAM_REDUCE_SUM));
        // the largest point gradient and the ranks' votes to stop (time limit): one MAX, only when termination is tested
        if (h->opt.check_termination) SOSLAM_CHECK(do_allreduce(h, h->scalp() + SC_GMAX_PTS, 2, SOSLAM_REDUCE_MAX));
        // acceptance test on the summed scalars (identical on every rank), then the publication
        published = ++h->publish_seq;
        launch_gate_publish(s, h->scalp(), h->x_cost, h->opt.min_relative_decrease, speculated ? 1 : 0, h->tail(), 4 + SC_COUNT, 4 + SC_LIN_ITERS,
                            4, h->host_raw, h->host_seq, published, is_constrained(h) ? 1 : 0);
    }
    if (speculated) {
        const double* gate = h->scalp() + SC_GATE;
        {
            StageScope sc(h, SOSLAM_STAGE_LINEARIZE);
            launch_linearize(s, h->n_tiles, h->tiles.p, h->uv.p, h->obs_pt.p, h->campre_c.p, h->pts[h->cur ^ 1].p, h->cam_free.p, h->proj,
                             h->opt.huber_delta, h->rows(), h->tile_part.p, gate);
        }
        if (!points_fused(h)) {
            StageScope sc(h, SOSLAM_STAGE_POINT_REDUCE);
            launch_point_reduce(s, h->n_pt, h->pt_start.p, h->pt_obs.p, h->q_cam.p, h->rows(), h->campre_c.p, h->C.p, h->gp.p, gate);
        }
    }
    {
        StageScope sc(h, SOSLAM_STAGE_SYNC);
        // the tail (this iteration's cost at the linearisation point, summed over ranks) lies right in front of the scalars
        // a kernel writes them into pinned host memory and then a sequence number the host polls: no copy command,
        // no completion signal between the GPU's last store and the host's decision
        SOSLAM_CHECK(wait_host_seq(h, h->host_seq, published));
    }
    SOSLAM_HIP_CHECK(hipGetLastError());
    // The band factor is exact, so one PCG round (= a direct solve plus the true residual) normally meets the
    // tolerance.  Rounds are enqueued without host checks: if this solve fell short - the step is still a valid
    // inexact step, its residual enters the model cost change - the next solve enqueues one round more; if it was
    // done before its last round, the next one enqueues only as many as were used.
    // The band part of a positive definite matrix need not be positive definite: if the factorisation of the truncated band
    // broke down, this problem goes back to block-Jacobi PCG for good (the step just taken is invalid and is repeated).
    if (h->off_band && h->use_cr && h->n_free && h->host_scal[SC_LIN_STATUS] == 1.0) {
        h->off_band = false; h->use_cr = false; h->pcg_band = false;
        h->bw = h->bw_full;
    }
    // With blocks outside the factored band (off_band) the factor is a preconditioner proper: more rounds, found the same way
    // (twice as many after a solve that fell short, as many as were used after one that did not need them all).
    if (h->use_cr && h->pcg_band && h->n_free && h->lagged_solve) {
        // rounds of the lagged solve: as many as it used, one more after a solve that fell short; a solve that needed more than
        // four sends the next iteration back to a fresh factor
        const int used = (int)h->host_scal[SC_LIN_ITERS];
        if (h->host_scal[SC_LIN_RESID] > h->opt.pcg_tolerance) { h->lag_rounds = std::min(h->lag_rounds + 1, 6); h->lag_ok = h->lag_rounds <= 4; }
        else h->lag_rounds = std::max(2, used);
    } else if (h->use_cr && h->pcg_band && h->n_free) {
        h->lag_ok = true;
        const int used = (int)h->host_scal[SC_LIN_ITERS];
        const int cap = h->off_band ? h->opt.pcg_max_iterations : 4;
        if (h->off_band) h->cr_rounds = std::max(2, std::min(used + 1, cap));   // the solve ran to its tolerance (take_step): one spare round
        else if (h->host_scal[SC_LIN_RESID] > h->opt.pcg_tolerance) h->cr_rounds = std::min(h->cr_rounds + 1, cap);
        else if (used >= 1 && used < h->cr_rounds) h->cr_rounds = used;
    }
    return SOSLAM_OK;
}

struct StepScalars {
    double x_cost, cand_cost, mcc, step_norm, x_norm, gdot, gmax;
    int lin_iters, lin_status, schur_status;
};

StepScalars read_scalars(const soslam_ba* h)
{
    const double* v = h->host_scal;
    StepScalars r;
    r.x_cost = h->host_raw[0];
    r.cand_cost = v[SC_CAND_COST];
    r.mcc = v[SC_MCC_PTS] + v[SC_MCC_CAM];
    r.step_norm = std::sqrt(v[SC_STEP2_PTS] + v[SC_STEP2_CAM]);
    r.x_norm = std::sqrt(v[SC_X2_PTS] + v[SC_X2_CAM]);
    r.gdot = v[SC_GDOT_PTS] + v[SC_GDOT_CAM];
    r.gmax = std::max(v[SC_GMAX_PTS], v[SC_GMAX_CAM]);
    r.lin_iters = (int)v[SC_LIN_ITERS];
    r.lin_status = (int)v[SC_LIN_STATUS];
    r.schur_status = (int)v[SC_SCHUR_STATUS];
    return r;
}

// ---- Ceres' line search on bounded problems -------------------------------------------------------------------------
// /root/reference/src/bundle_adjuster.cpp:104-108 bounds every point coordinate, so Ceres treats the problem as constrained:
// every valid trust-region step goes through TrustRegionMinimizer::DoLineSearch before its candidate is evaluated - an
// Armijo search (sufficient decrease 1e-4) from step size 1 along the step, trial points projected onto the box, CUBIC
// interpolation on cost and directional derivative (both evaluated at every trial), contraction within [1e-3, 0.6] of the
// last step size, at most 20 iterations, step sizes below 1e-9 / |delta|_inf given up; on success the step is scaled by
// the size found, on failure it is left alone; the model cost change stays that of the full step.  The host runs the
// search (a handful of scalars per trial), the device evaluates the trials (ba_ls_candidate / ba_ls_eval / ba_ls_sums).
// The first trial - step size 1 - is the candidate the iteration has already evaluated, so the search costs nothing
// unless that candidate fails the sufficient-decrease test.
struct LsSample { double x = 0.0, value = 0.0, gradient = 0.0; bool value_ok = false, gradient_ok = false; };

double ls_polyval(const std::vector<double>& c, double x)   // c[0] x^n + ... + c[n]
{
    double v = 0.0;
    for (double ci : c) v = v * x + ci;
    return v;
}

// polynomial through the samples' values and gradients (Ceres FindInterpolatingPolynomial), highest power first
std::vector<double> ls_interpolate(const std::vector<LsSample>& smp)
{
    int nc = 0;
    for (const LsSample& q : smp) nc += (q.value_ok ? 1 : 0) + (q.gradient_ok ? 1 : 0);
    const int deg = nc - 1;
    std::vector<std::vector<double>> a((size_t)nc, std::vector<double>((size_t)nc + 1, 0.0));
    int row = 0;
    for (const LsSample& q : smp) {
        if (q.value_ok) {
            for (int j = 0; j <= deg; j++) a[(size_t)row][(size_t)j] = std::pow(q.x, deg - j);
            a[(size_t)row][(size_t)nc] = q.value; row++;
        }
        if (q.gradient_ok) {
            for (int j = 0; j < deg; j++) a[(size_t)row][(size_t)j] = (deg - j) * std::pow(q.x, deg - j - 1);
            a[(size_t)row][(size_t)nc] = q.gradient; row++;
        }
    }
    for (int k = 0; k < nc; k++) {   // elimination with row pivoting
        int piv = k;
        for (int r = k + 1; r < nc; r++) if (std::fabs(a[(size_t)r][(size_t)k]) > std::fabs(a[(size_t)piv][(size_t)k])) piv = r;
        std::swap(a[(size_t)k], a[(size_t)piv]);
        if (a[(size_t)k][(size_t)k] == 0.0) continue;
        for (int r = k + 1; r < nc; r++) {
            const double f = a[(size_t)r][(size_t)k] / a[(size_t)k][(size_t)k];
            for (int j = k; j <= nc; j++) a[(size_t)r][(size_t)j] -= f * a[(size_t)k][(size_t)j];
        }
    }
    std::vector<double> c((size_t)nc, 0.0);
    for (int k = nc - 1; k >= 0; k--) {
        double v = a[(size_t)k][(size_t)nc];
        for (int j = k + 1; j < nc; j++) v -= a[(size_t)k][(size_t)j] * c[(size_t)j];
        c[(size_t)k] = a[(size_t)k][(size_t)k] != 0.0 ? v / a[(size_t)k][(size_t)k] : 0.0;
    }
    return c;
}

// real parts of all roots (Ceres evaluates the polynomial at the real part of complex roots too)
std::vector<double> ls_root_real_parts(std::vector<double> c)
{
    while (c.size() > 1 && c.front() == 0.0) c.erase(c.begin());
    const int deg = (int)c.size() - 1;
    if (deg <= 0) return {};
    if (deg == 1) return {-c[1] / c[0]};
    if (deg == 2) {
        const double a = c[0], b = c[1], cc = c[2], D = b * b - 4 * a * cc, sD = std::sqrt(std::fabs(D));
        if (D >= 0) return b >= 0 ? std::vector<double>{(-b - sD) / (2.0 * a), (2.0 * cc) / (-b - sD)}
                                  : std::vector<double>{(2.0 * cc) / (-b + sD), (-b + sD) / (2.0 * a)};
        return {-b / (2.0 * a), -b / (2.0 * a)};
    }
    // Aberth-Ehrlich iteration on the monic polynomial
    using cd = std::complex<double>;
    std::vector<double> m((size_t)deg + 1);
    for (int i = 0; i <= deg; i++) m[(size_t)i] = c[(size_t)i] / c[0];
    double rad = 0.0;
    for (int i = 1; i <= deg; i++) rad = std::max(rad, std::pow(std::fabs(m[(size_t)i]), 1.0 / i));
    if (rad == 0.0) return std::vector<double>((size_t)deg, 0.0);
    std::vector<cd> z((size_t)deg);
    for (int i = 0; i < deg; i++) z[(size_t)i] = std::polar(rad, 2.0 * 3.14159265358979323846 * i / deg + 0.4);
    for (int it = 0; it < 200; it++) {
        double worst = 0.0;
        for (int i = 0; i < deg; i++) {
            cd pv(1.0, 0.0), dv(0.0, 0.0);
            for (int k = 1; k <= deg; k++) { dv = dv * z[(size_t)i] + pv; pv = pv * z[(size_t)i] + m[(size_t)k]; }
            if (dv == cd(0.0, 0.0)) continue;
            const cd w = pv / dv;
            cd sum(0.0, 0.0);
            for (int j = 0; j < deg; j++) if (j != i && z[(size_t)i] != z[(size_t)j]) sum += 1.0 / (z[(size_t)i] - z[(size_t)j]);
            const cd q = 1.0 - w * sum;
            const cd corr = q == cd(0.0, 0.0) ? w : w / q;
            z[(size_t)i] -= corr;
            worst = std::max(worst, std::abs(corr));
        }
        if (worst <= 1e-15 * rad) break;
    }
    std::vector<double> re;
    for (const cd& q : z) re.push_back(q.real());
    return re;
}

// Ceres MinimizePolynomial on [x_min, x_max]: midpoint, both ends, the derivative's roots inside
double ls_minimize(const std::vector<double>& c, double x_min, double x_max)
{
    double best_x = 0.5 * (x_min + x_max), best = ls_polyval(c, best_x);
    for (double x : {x_min, x_max}) { const double v = ls_polyval(c, x); if (v < best) { best = v; best_x = x; } }
    const int deg = (int)c.size() - 1;
    if (deg <= 1) return best_x;
    std::vector<double> d((size_t)deg);
    for (int i = 0; i < deg; i++) d[(size_t)i] = (deg - i) * c[(size_t)i];
    for (double r : ls_root_real_parts(d)) {
        if (r < x_min || r > x_max) continue;
        const double v = ls_polyval(c, r);
        if (v < best) { best = v; best_x = r; }
    }
    return best_x;
}

// one trial of the search: candidate x+ = Plus(x, a delta) into cams[cur ^ 1] / pts[cur ^ 1] / campre_c, its cost,
// direction . gradient, |x+ - x|^2 and |delta|_inf (summed / maximised over ranks) into h->host_scal[SC_LS_*]
int ls_evaluate(soslam_ba* h, double a)
{
    hipStream_t s = h->stream;
    if (points_only(h) && h->points_only_ready) {
        // structure only: the trial in one launch (the cameras and both pose tables are the constant ones already)
        PointsStepArgs pa{};
        pa.n_pt = h->n_pt; pa.pt_start = h->pt_start.p; pa.pt_obs = h->pt_obs.p; pa.q_cam = h->q_cam.p; pa.uv = h->uv.p;
        pa.campre = h->campre.p; pa.pts = h->pts[h->cur].p; pa.pts_out = h->pts[h->cur ^ 1].p; pa.dp = h->dp.p;
        pa.huber_delta = h->opt.huber_delta; pa.bound_lo = h->opt.lower_bound; pa.bound_hi = h->opt.upper_bound; pa.scal = h->scalp();
        const unsigned long long seq = ++h->publish_seq;
        launch_points_ls(s, pa, h->proj, a, h->tail(), 4 + SC_COUNT, h->host_raw, h->host_seq, seq);
        SOSLAM_HIP_CHECK(hipGetLastError());
        return wait_host_seq(h, h->host_seq, seq);
    }
    SOSLAM_CHECK(h->ls_tile.alloc(2 * (size_t)std::max<uint32_t>(h->n_tiles, 1)));
    SOSLAM_CHECK(h->ls_part.alloc(2 * (size_t)ls_candidate_blocks(h->n_pt)));
    launch_ls_candidate(s, h->n_cam, h->n_pt, h->cams[h->cur].p, h->pts[h->cur].p, h->dc_full.p, h->dp.p, a, h->opt.lower_bound,
                        h->opt.upper_bound, h->cams[h->cur ^ 1].p, h->pts[h->cur ^ 1].p, h->ls_part.p, h->campre_c.p);
    launch_ls_eval(s, h->n_tiles, h->tiles.p, h->uv.p, h->obs_pt.p, h->campre_c.p, h->pts[h->cur ^ 1].p, h->dc_full.p, h->dp.p, h->cam_free.p,
                   h->proj, h->opt.huber_delta, h->ls_tile.p);
    const unsigned long long seq = ++h->publish_seq;
    if (!h->collective()) {
        launch_ls_sums(s, h->ls_tile.p, h->n_tiles, h->ls_part.p, ls_candidate_blocks(h->n_pt), h->scalp(), h->tail(), 4 + SC_COUNT, h->host_raw,
                       h->host_seq, seq);
    } else {
        launch_ls_sums(s, h->ls_tile.p, h->n_tiles, h->ls_part.p, ls_candidate_blocks(h->n_pt), h->scalp(), nullptr, 0, nullptr, nullptr, 0);
        SOSLAM_CHECK(do_allreduce(h, h->scalp() + SC_LS_COST, 3, SOSLAM_REDUCE_SUM));
        SOSLAM_CHECK(do_allreduce(h, h->scalp() + SC_LS_DMAX, 1, SOSLAM_REDUCE_MAX));
        launch_publish(s, h->tail(), 4 + SC_COUNT, 4 + SC_LIN_ITERS, 0, h->host_raw, h->host_seq, seq);
    }
    return wait_host_seq(h, h->host_seq, seq);
}

// ArmijoLineSearch::DoSearch from step size 1 (whose cost is already known).  Out: step size (1 = delta untouched), and when
// it is < 1 the accepted trial's cost and |x+ - x| (the candidate buffers then hold that trial).
int line_search(soslam_ba* h, const StepScalars& sc, double* step_size, double* cand_cost, double* step_norm, int* n_iterations)
{
    constexpr double kSufficientDecrease = 1e-4, kMaxContraction = 1e-3, kMinContraction = 0.6, kMinStepSize = 1e-9;
    constexpr int kMaxIterations = 20;
    *step_size = 1.0;
    if (!is_constrained(h)) return SOSLAM_OK;                                          // unconstrained problem: Ceres does not search
    if (sc.cand_cost <= sc.x_cost + kSufficientDecrease * sc.gdot) return SOSLAM_OK;   // the full step satisfies Armijo
    const double step2_cam_full = h->host_scal[SC_STEP2_CAM];
    LsSample initial, previous, current;
    initial.x = 0.0; initial.value = sc.x_cost; initial.gradient = sc.gdot; initial.value_ok = initial.gradient_ok = true;
    auto evaluate = [&](double a, LsSample& out) -> int {
        SOSLAM_CHECK(ls_evaluate(h, a));
        out = LsSample{};
        out.x = a;
        out.value = h->host_scal[SC_LS_COST];
        out.gradient = h->host_scal[SC_LS_DIR];
        out.value_ok = std::isfinite(out.value);
        out.gradient_ok = out.value_ok && std::isfinite(out.gradient);
        return SOSLAM_OK;
    };
    // Ceres evaluates cost AND gradient at the first trial (CUBIC interpolation).  The structure-only step kernel has left
    // both in the line-search slots already (its candidate IS the trial at step size 1)
    if (points_only(h) && h->points_only_ready) {
        current = LsSample{};
        current.x = 1.0;
        current.value = h->host_scal[SC_LS_COST];
        current.gradient = h->host_scal[SC_LS_DIR];
        current.value_ok = std::isfinite(current.value);
        current.gradient_ok = current.value_ok && std::isfinite(current.gradient);
    } else {
        SOSLAM_CHECK(evaluate(1.0, current));
    }
    const double dmax = h->host_scal[SC_LS_DMAX];
    int iters = 0;
    bool found = true;
    while (!current.value_ok || current.value > sc.x_cost + kSufficientDecrease * sc.gdot * current.x) {
        iters++;
        if (iters >= kMaxIterations) { found = false; break; }
        double step;
        const double lo = kMaxContraction * current.x, hi = kMinContraction * current.x;
        if (!current.value_ok) {
            step = std::min(std::max(current.x * 0.5, lo), hi);
        } else {
            std::vector<LsSample> smp{initial, current};
            if (previous.value_ok) smp.push_back(previous);
            step = ls_minimize(ls_interpolate(smp), lo, hi);
        }
        if (step * dmax < kMinStepSize) { found = false; break; }
        previous = current;
        SOSLAM_CHECK(evaluate(step, current));
    }
    *n_iterations += iters;
    if (!found) {
        // the search failed: Ceres leaves delta alone - the candidate buffers must hold the full step again
        SOSLAM_CHECK(evaluate(1.0, current));
        return SOSLAM_OK;
    }
    *step_size = current.x;
    *cand_cost = current.value;
    *step_norm = std::sqrt(h->host_scal[SC_LS_STEP2] + current.x * current.x * step2_cam_full);
    return SOSLAM_OK;
}

// Structure-only problems (the per-frame call, /root/reference/src/slam.cpp:123): the whole loop below in ONE launch, controller and
// line search included (ba_points.hip: ba_points_solve).  The host enqueues it, polls one sequence word and copies the record
// and the iteration log out of pinned memory.  Verbose and per-stage profiling runs keep the host loop (one launch per
// iteration and per trial), as does SOSLAM_NO_RESIDENT_SOLVE=1.
bool resident_solve(const soslam_ba* h, int max_it)
{
    static const bool off = [] { const char* e = getenv("SOSLAM_NO_RESIDENT_SOLVE"); return e && *e && *e != '0'; }();
    return !off && points_only(h) && max_it > 0 && !h->opt.verbose && !h->opt.profile_stages;
}

static_assert(sizeof(soslam_ba_iteration) == sizeof(double) * kPointsLogDoubles, "the device writes the log in this layout");
static_assert(kPointsTermMaxIterations == SOSLAM_TERM_MAX_ITERATIONS && kPointsTermParameter == SOSLAM_TERM_PARAMETER_TOLERANCE &&
              kPointsTermFunction == SOSLAM_TERM_FUNCTION_TOLERANCE && kPointsTermGradient == SOSLAM_TERM_GRADIENT_TOLERANCE &&
              kPointsTermMinRadius == SOSLAM_TERM_MIN_RADIUS && kPointsTermInvalid == SOSLAM_TERM_INVALID_STEPS &&
              kPointsTermTime == SOSLAM_TERM_TIME, "termination codes of the device controller");

int run_points_resident(soslam_ba* h, int max_it, bool check, soslam_ba_summary* out)
{
    const soslam_ba_options& o = h->opt;
    hipStream_t s = h->stream;
    soslam_ba_summary sum{};
    sum.linear_solver = h->solver;
    sum.setup_seconds = h->setup_seconds;
    const double t0 = now_sec();
    h->log.clear();
    const int entries = max_it + 1;
    constexpr int kHead = PSV_COUNT + 1;   // record, sequence word
    if (entries > h->ps_host_entries) {
        h->pinned_free(h->ps_host);
        h->ps_host = nullptr; h->ps_host_entries = 0;
        const int cap = std::max(entries, 64);
        SOSLAM_CHECK(h->pinned_alloc(reinterpret_cast<void**>(&h->ps_host), sizeof(double) * ((size_t)kHead + (size_t)cap * kPointsLogDoubles)));
        h->ps_host_entries = cap;
        *reinterpret_cast<unsigned long long*>(h->ps_host + PSV_COUNT) = 0;
    }
    SOSLAM_CHECK(h->ps_log.alloc((size_t)h->ps_host_entries * kPointsLogDoubles));
    // the kernel needs the constant cameras' pose table and nothing else (the candidate table, the second camera buffer and the zero
    // camera step belong to the host-driven loop, which prepares them itself: points_only_ready)
    if (!h->points_only_ready && !h->points_resident_ready) {
        launch_pose_prepare(s, h->n_cam, h->cams[h->cur].p, h->campre.p);
        h->points_resident_ready = true;
    }
    PointsStepArgs a{};
    a.n_pt = h->n_pt; a.pt_start = h->pt_start.p; a.pt_obs = h->pt_obs.p; a.q_cam = h->q_cam.p; a.uv = h->uv.p;
    a.campre = h->campre.p; a.pts = h->pts[h->cur].p; a.pts_out = h->pts[h->cur ^ 1].p; a.dp = h->dp.p;
    a.C = h->C.p; a.gp = h->gp.p; a.sp = h->sp.p; a.Cinv = h->Cinv.p; a.huber_delta = o.huber_delta;
    a.bound_lo = o.lower_bound; a.bound_hi = o.upper_bound; a.jacobi = o.jacobi_scaling; a.scal = h->scalp(); a.cost_x_out = h->tail();
    PointsSolveCtl c{};
    c.pts[0] = h->pts[0].p; c.pts[1] = h->pts[1].p;
    c.cur = h->cur; c.invalid_run = h->invalid_run; c.init_scale = h->scale_init ? 0 : 1; c.x_cost_known = h->x_cost_known ? 1 : 0;
    c.radius = h->radius; c.decrease_factor = h->decrease_factor; c.x_cost = h->x_cost;
    c.max_it = max_it; c.check = check ? 1 : 0; c.constrained = is_constrained(h) ? 1 : 0;
    c.lm_lo = o.min_lm_diagonal; c.lm_hi = o.max_lm_diagonal;
    c.min_radius = o.min_radius; c.max_radius = o.max_radius; c.min_relative_decrease = o.min_relative_decrease;
    c.gradient_tolerance = o.gradient_tolerance; c.parameter_tolerance = o.parameter_tolerance; c.function_tolerance = o.function_tolerance;
    c.max_ticks = (check && o.max_solver_time_seconds > 0.0) ? std::max<long long>(1, (long long)(o.max_solver_time_seconds * 1e8)) : 0;
    c.log = h->ps_log.p;
    constexpr size_t kSyncRecords = 2 * (size_t)(kPointsOnlyMax / 64) * 16;
    if (!h->ps_sync.p) {
        SOSLAM_CHECK(h->ps_sync.alloc(kSyncRecords + 16));
        SOSLAM_CHECK(h->ps_sync.zero(s));
        h->ps_sync_base = 0;
    }
    c.n_wg = (int)((h->n_pt + 63) / 64);
    c.sync_records = h->ps_sync.p;
    c.sync_counter = reinterpret_cast<unsigned long long*>(h->ps_sync.p + kSyncRecords);
    c.sync_base = h->ps_sync_base;
    c.host_record = h->ps_host;
    c.host_seq = reinterpret_cast<unsigned long long*>(h->ps_host + PSV_COUNT);
    c.host_log = h->ps_host + kHead;
    c.seq = ++h->publish_seq;
    h->scale_init = true;
    launch_points_solve(s, a, h->proj, c);
    if (const int rc = wait_host_seq(h, c.host_seq, c.seq)) {
        // the record never came (a failed launch, a workgroup that never ran): the barrier's counter is unknown from here on
        (void)hipStreamSynchronize(s);
        (void)h->ps_sync.zero(s);
        h->ps_sync_base = 0;
        return rc;
    }
    const double* r = h->ps_host;
    h->ps_sync_base += (unsigned long long)r[PSV_PASSES] * (unsigned long long)c.n_wg;
    if (r[PSV_ERROR] == 2.0) {
        // a workgroup never arrived at the grid barrier: the counter is no longer what the host thinks it is
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        SOSLAM_CHECK(h->ps_sync.zero(s));
        h->ps_sync_base = 0;
        set_last_error("the structure-only solve's grid barrier timed out");
        return SOSLAM_ERR_HIP;
    }
    if (r[PSV_ERROR] != 0.0) {
        set_last_error("non-finite cost at the initial point");
        return SOSLAM_ERR_NON_FINITE;
    }
    h->radius = r[PSV_RADIUS]; h->decrease_factor = r[PSV_DECREASE]; h->x_cost = r[PSV_X_COST]; h->x_cost_known = true;
    h->cur = (int)r[PSV_CUR]; h->invalid_run = (int)r[PSV_INVALID_RUN];
    h->linearized = true; h->campre_current = false;
    const int n_log = (int)r[PSV_N_LOG];
    h->log.resize((size_t)n_log);
    if (n_log) std::memcpy(h->log.data(), h->ps_host + kHead, sizeof(soslam_ba_iteration) * (size_t)n_log);
    sum.iterations = (int)r[PSV_ITERATIONS];
    sum.accepted = (int)r[PSV_ACCEPTED];
    sum.termination = (int)r[PSV_TERMINATION];
    sum.line_search_steps = (int)r[PSV_LS_STEPS];
    sum.initial_cost = r[PSV_INITIAL_COST];
    sum.final_cost = h->x_cost;
    sum.solve_seconds = now_sec() - t0;
    if (out) *out = sum;
    return SOSLAM_OK;
}

// The Levenberg-Marquardt loop.  fixed_count >= 0: exactly that many iterations, no termination tests.
int run_lm(soslam_ba* h, int fixed_count, soslam_ba_summary* out)
{
    if (!h->have_problem || !h->have_state || !h->have_proj) {
        set_last_error("solve called before set_projection / set_problem / set_state");
        return SOSLAM_ERR_STATE;
    }
    const soslam_ba_options& o = h->opt;
    const bool check = fixed_count < 0 && o.check_termination;
    const int max_it = fixed_count >= 0 ? fixed_count : o.max_iterations;
    if (resident_solve(h, max_it)) return run_points_resident(h, max_it, check, out);
    soslam_ba_summary sum{};
    sum.linear_solver = h->solver;
    sum.setup_seconds = h->setup_seconds;
    sum.termination = SOSLAM_TERM_MAX_ITERATIONS;
    const double t0 = now_sec();
    h->log.clear();
    h->stop_agreed = false;
    hipStream_t s = h->stream;

    if (!h->linearized) SOSLAM_CHECK(linearize(h, true));
    // the solver / elimination status words start clean; every iteration's publication clears them again
    static_assert(SC_SCHUR_STATUS == SC_LIN_ITERS + 3, "status slots are contiguous");
    SOSLAM_HIP_CHECK(hipMemsetAsync(h->scalp() + SC_LIN_ITERS, 0, 4 * sizeof(double), s));
    bool have_initial = false;
    int it = 0;
    while (true) {
        if (it >= max_it) { sum.termination = SOSLAM_TERM_MAX_ITERATIONS; break; }
        if (check && h->radius < o.min_radius) { sum.termination = SOSLAM_TERM_MIN_RADIUS; break; }
        // The wall-clock test (/root/reference/src/params.h:41 through bundle_adjuster.cpp:18) reads this rank's clock.  On one
        // rank it ends the loop here, as Ceres does at the top of an iteration.  In a multi-rank job a rank-local exit
        // would leave the other ranks waiting in the next all-reduce, so the rank VOTES: the vote travels with this
        // iteration's MAX all-reduce and every rank leaves together at the top of the next one.
        bool stop_vote = false;
        if (check && o.max_solver_time_seconds > 0.0 && now_sec() - t0 > o.max_solver_time_seconds) {
            if (!h->collective()) { sum.termination = SOSLAM_TERM_TIME; break; }
            stop_vote = true;
        }
        if (h->stop_agreed) { h->stop_agreed = false; sum.termination = SOSLAM_TERM_TIME; break; }
        if (!h->linearized) SOSLAM_CHECK(linearize(h, true));   // only after a speculative linearisation the host did not follow
        const double radius = h->radius;
        SOSLAM_CHECK(take_step(h, radius, true, stop_vote));
        StepScalars sc = read_scalars(h);
        if (check && h->collective() && h->host_scal[SC_STOP] != 0.0) h->stop_agreed = true;
        // the device accepted the step and has already linearised at the candidate (speculation, see take_step); if the
        // host ends up keeping x after all, the linearisation at x has to be made again
        bool dev_linearized = h->host_scal[SC_GATE] != 0.0;
        struct Relinearize {
            soslam_ba* h; bool* armed;
            ~Relinearize() { if (*armed) { h->linearized = false; } }
        } relin{h, &dev_linearized};
        if (h->x_cost_known) sc.x_cost = h->x_cost;   // the device slot is stale then (see linearize)
        h->x_cost = sc.x_cost;
        h->x_cost_known = true;
        if (!have_initial) {
            have_initial = true;
            sum.initial_cost = sc.x_cost;
            if (!std::isfinite(sc.x_cost)) {
                set_last_error("non-finite cost at the initial point");
                return SOSLAM_ERR_NON_FINITE;
            }
            soslam_ba_iteration e{};
            e.cost = sc.x_cost; e.radius = radius; e.gradient_max_norm = sc.gmax; e.accepted = 1; e.valid = 1;
            h->log.push_back(e);
        } else if (!h->log.empty() && h->log.back().accepted) {
            h->log.back().gradient_max_norm = sc.gmax;  // gradient at the point the last accepted step reached
        }
        if (check && sc.gmax <= o.gradient_tolerance) { sum.termination = SOSLAM_TERM_GRADIENT_TOLERANCE; break; }
        it++;
        soslam_ba_iteration e{};
        e.cost = sc.x_cost; e.radius = radius; e.gradient_max_norm = sc.gmax;
        e.model_cost_change = sc.mcc; e.linear_iterations = sc.lin_iters;
        sum.linear_iterations += sc.lin_iters;
        // a non-finite candidate cost is how another rank's failed point elimination arrives here (launch_status_poison)
        const bool lin_ok = sc.schur_status == 0 && sc.lin_status == 0 && std::isfinite(sc.mcc) && std::isfinite(sc.cand_cost);
        if (!lin_ok || !(sc.mcc > 0.0)) {
            e.valid = 0;
            h->log.push_back(e);
            h->last_rel_decrease = 1.0;
            h->cr_factor_valid = false;   // whatever broke this solve must not precondition the next one
            if (check && ++h->invalid_run >= 5) { sum.termination = SOSLAM_TERM_INVALID_STEPS; break; }
            // Ceres: TrustRegionMinimizer::HandleInvalidStep -> LevenbergMarquardtStrategy::StepIsInvalid halves the radius
            // and leaves the rejected-step factor alone (a later rejection still divides by the factor it would have used)
            h->radius *= 0.5;
            if (o.verbose)
                printf("[rank %d] %4d  cost %.9e  invalid step (model change %.3e, candidate %.9e, solver status %d, elimination status %d, lin_it %d, "
                       "lin_res %.2e)  radius %.3e\n", h->rank, it, sc.x_cost, sc.mcc, sc.cand_cost, sc.lin_status, sc.schur_status, sc.lin_iters,
                       h->host_scal[SC_LIN_RESID], radius);
            continue;
        }
        h->invalid_run = 0;
        e.valid = 1;
        // bounded problem: the line search along the projected step, before the candidate is judged (see line_search)
        {
            double a = 1.0, ls_cost = 0.0, ls_norm = 0.0;
            int ls_iters = 0;
            SOSLAM_CHECK(line_search(h, sc, &a, &ls_cost, &ls_norm, &ls_iters));
            sum.line_search_steps += ls_iters;
            if (a != 1.0) { sc.cand_cost = ls_cost; sc.step_norm = ls_norm; }
        }
        e.candidate_cost = sc.cand_cost;
        e.step_norm = sc.step_norm;
        if (check) {
            if (sc.step_norm <= o.parameter_tolerance * (sc.x_norm + o.parameter_tolerance)) {
                h->log.push_back(e);
                sum.termination = SOSLAM_TERM_PARAMETER_TOLERANCE;
                break;
            }
            if (std::fabs(sc.x_cost - sc.cand_cost) <= o.function_tolerance * sc.x_cost) {
                h->log.push_back(e);
                sum.termination = SOSLAM_TERM_FUNCTION_TOLERANCE;
                break;
            }
        }
        const double rel = (sc.x_cost - sc.cand_cost) / sc.mcc;
        e.relative_decrease = rel;
        h->last_rel_decrease = 1.0;   // a rejected step: the next solve gets a fresh factor (take_step: lagged factor)
        if (rel > o.min_relative_decrease) {
            h->last_rel_decrease = sc.x_cost > 0.0 ? (sc.x_cost - sc.cand_cost) / sc.x_cost : 1.0;
            h->cur ^= 1;
            h->campre.swap(h->campre_c);   // the candidate's table becomes the linearisation point's
            h->campre_current = true;
            h->x_cost = sc.cand_cost;
            h->x_cost_known = true;
            if (dev_linearized) {
                if (points_fused(h)) h->pt_blocks_valid = false;   // no point pass behind the gate: the Schur kernel forms C, g_p
                dev_linearized = false;      // consumed: B, g_c, C, g_p and the compact rows are those of the new point
                h->campre_current = false;
                h->linearized = true;
            } else {
                SOSLAM_CHECK(linearize(h, true));
            }
            double f = 1.0 - std::pow(2.0 * rel - 1.0, 3.0);
            if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            h->radius = std::min(o.max_radius, h->radius / f);
            h->decrease_factor = 2.0;
            e.accepted = 1;
            e.cost = sc.cand_cost;
            sum.accepted++;
        } else {
            h->radius /= h->decrease_factor;
            h->decrease_factor *= 2.0;
        }
        h->log.push_back(e);
        if (o.verbose)
            printf("[rank %d] %4d  cost %.9e  cand %.9e  model %.3e  rho %.3e  |step| %.3e  radius %.3e  lin_it %d  lin_res %.2e  %s\n", h->rank, it, sc.x_cost,
                   sc.cand_cost, sc.mcc, rel, sc.step_norm, radius, sc.lin_iters, h->host_scal[SC_LIN_RESID], e.accepted ? "accepted" : "rejected");
    }
    SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
    sum.iterations = it;
    sum.final_cost = h->x_cost;
    if (!have_initial) {
        // zero iterations requested (or the clock ran out before the first): report the cost of the current point
        if (!h->x_cost_known && points_only(h)) {
            // the structure-only path has no linearisation of its own: one step kernel, of which only the cost at x is kept
            SOSLAM_CHECK(take_step(h, h->radius));
            h->x_cost = h->host_raw[0];
        } else if (!h->x_cost_known) {
            SOSLAM_HIP_CHECK(hipMemcpyAsync(h->host_scal, h->scalp(), sizeof(double), hipMemcpyDeviceToHost, s));
            SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
            h->x_cost = h->host_scal[0];
        }
        sum.initial_cost = sum.final_cost = h->x_cost;
    }
    sum.solve_seconds = now_sec() - t0;
    collect_stage_times(h, &sum);
    if (out) *out = sum;
    return SOSLAM_OK;
}

// pinned staging of the state transfers (region 2): a two-entry table in front, `bytes` of payload behind it
int state_staging(soslam_ba* h, size_t bytes, double** payload)
{
    constexpr size_t kTable = 64;   // two PackedSeg, padded
    static_assert(2 * sizeof(PackedSeg) <= kTable, "table room");
    const size_t need = kTable + bytes;
    if (need > h->up_host_bytes[2]) {
        h->pinned_free(h->up_host[2]);
        h->up_host[2] = nullptr; h->up_host_bytes[2] = 0;
        const size_t cap = std::max<size_t>(need + need / 4, 1u << 16);
        SOSLAM_CHECK(h->pinned_alloc(reinterpret_cast<void**>(&h->up_host[2]), cap));
        h->up_host_bytes[2] = cap;
    }
    *payload = reinterpret_cast<double*>(h->up_host[2] + kTable);
    return SOSLAM_OK;
}

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------

extern "C" {

const char* soslam_version(void) { return "soslam_ba 0.1.0 (gfx950)"; }

const char* soslam_status_string(int status)
{
    switch (status) {
    case SOSLAM_OK: return "ok";
    case SOSLAM_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SOSLAM_ERR_HIP: return "HIP runtime error";
    case SOSLAM_ERR_NO_DEVICE: return "no usable gfx950 device";
    case SOSLAM_ERR_NON_FINITE: return "non-finite cost";
    case SOSLAM_ERR_LINEAR_SOLVER: return "linear solver failure";
    case SOSLAM_ERR_COMM: return "all-reduce failed";
    case SOSLAM_ERR_STATE: return "call sequence error";
    default: return "unknown status";
    }
}

const char* soslam_last_error(void) { return g_last_error.c_str(); }

void soslam_ba_options_default(soslam_ba_options* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->max_iterations = 50;
    o->check_termination = 1;
    o->linear_solver = SOSLAM_SOLVER_AUTO;
    o->pcg_max_iterations = 500;
    o->pcg_tolerance = 1e-10;
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
    o->max_solver_time_seconds = 0.0;
    o->jacobi_scaling = 1;
    o->verbose = 0;
    o->device = -1;
    o->profile_stages = 0;
    o->stream = nullptr;
}

int soslam_ba_create(const soslam_ba_options* opts, soslam_ba** out)
{
    if (!out) return SOSLAM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        set_last_error("no HIP device visible: the bundle-adjustment backend has no CPU fallback");
        return SOSLAM_ERR_NO_DEVICE;
    }
    std::unique_ptr<soslam_ba> h(new soslam_ba());
    if (opts) h->opt = *opts; else soslam_ba_options_default(&h->opt);
    if (h->opt.device >= 0) {
        if (h->opt.device >= n_dev) { set_last_error("device %d out of range (%d visible)", h->opt.device, n_dev); return SOSLAM_ERR_INVALID_ARGUMENT; }
        SOSLAM_HIP_CHECK(hipSetDevice(h->opt.device));
    }
    SOSLAM_HIP_CHECK(hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    SOSLAM_HIP_CHECK(hipGetDeviceProperties(&prop, h->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error("device %d is %s; this library carries gfx950 code objects only", h->device, prop.gcnArchName);
        return SOSLAM_ERR_NO_DEVICE;
    }
    if (h->opt.stream) {
        h->stream = static_cast<hipStream_t>(h->opt.stream);
    } else {
        SOSLAM_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    *out = h.release();
    return SOSLAM_OK;
}

int soslam_ba_set_options(soslam_ba* h, const soslam_ba_options* opts)
{
    if (!h || !opts) return SOSLAM_ERR_INVALID_ARGUMENT;
    if ((opts->device >= 0 && opts->device != h->device) || (opts->stream && opts->stream != h->opt.stream)) {
        set_last_error("set_options cannot move a handle to another device or stream");
        return SOSLAM_ERR_INVALID_ARGUMENT;
    }
    void* const stream = h->opt.stream;
    const int32_t device = h->opt.device;
    h->opt = *opts;
    h->opt.stream = stream;
    h->opt.device = device;
    h->linearized = false; h->x_cost_known = false;   // the linear solver was chosen (and its work space sized) in set_problem and stays
    return SOSLAM_OK;
}

void soslam_ba_destroy(soslam_ba* h)
{
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    delete h;
}

int soslam_ba_set_projection(soslam_ba* h, const double* pl, const double* pr)
{
    if (!h || !pl || !pr) return SOSLAM_ERR_INVALID_ARGUMENT;
    std::memcpy(h->proj.l, pl, sizeof h->proj.l);
    std::memcpy(h->proj.r, pr, sizeof h->proj.r);
    h->have_proj = true;
    h->linearized = false; h->x_cost_known = false;
    return SOSLAM_OK;
}

int soslam_ba_set_problem(soslam_ba* h, uint32_t n_cam, uint32_t n_pt, uint32_t n_obs, const uint32_t* obs_cam,
                          const uint32_t* obs_pt, const float* obs_uv, const uint8_t* cam_fixed)
{
    if (!h || n_cam == 0 || (n_obs && (!obs_cam || !obs_pt || !obs_uv))) return SOSLAM_ERR_INVALID_ARGUMENT;
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    h->have_problem = false;
    h->linearized = false; h->x_cost_known = false;
    return build_problem(h, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_uv, cam_fixed);
}

int soslam_ba_set_state(soslam_ba* h, const double* poses, const double* points)
{
    if (!h || !poses || (h->n_pt && !points)) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_problem) { set_last_error("set_state before set_problem"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    h->cur = 0;
    const size_t cam_bytes = sizeof(double) * 6 * h->n_cam, pt_bytes = sizeof(double) * 3 * (size_t)h->n_pt;
    if (cam_bytes + pt_bytes <= kPackedUploadMax && !getenv("SOSLAM_NO_PACKED_UPLOAD")) {
        // small states: permuted straight into pinned staging, one kernel moves both arrays
        double* stage = nullptr;
        SOSLAM_CHECK(state_staging(h, cam_bytes + pt_bytes, &stage));
        PackedSeg* table = reinterpret_cast<PackedSeg*>(h->up_host[2]);
        std::memcpy(stage, poses, cam_bytes);
        double* sp = stage + 6 * (size_t)h->n_cam;
        for (uint32_t i = 0; i < h->n_pt; i++) std::memcpy(sp + 3 * (size_t)i, points + 3 * (size_t)h->pt_int2user[i], 3 * sizeof(double));
        table[0] = PackedSeg{h->cams[0].p, 0, cam_bytes};
        table[1] = PackedSeg{h->pts[0].p, cam_bytes, pt_bytes};
        launch_packed_scatter(h->stream, table, h->n_pt ? 2 : 1, reinterpret_cast<const unsigned char*>(stage));
    } else {
        std::vector<double> p((size_t)h->n_pt * 3);
        for (uint32_t i = 0; i < h->n_pt; i++) std::memcpy(&p[3 * (size_t)i], points + 3 * (size_t)h->pt_int2user[i], 3 * sizeof(double));
        SOSLAM_HIP_CHECK(hipMemcpyAsync(h->cams[0].p, poses, cam_bytes, hipMemcpyHostToDevice, h->stream));
        if (h->n_pt) SOSLAM_HIP_CHECK(hipMemcpyAsync(h->pts[0].p, p.data(), pt_bytes, hipMemcpyHostToDevice, h->stream));
    }
    SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->last_rel_decrease = 1.0;   // a new starting point: the first solve factors afresh
    h->radius = h->opt.initial_radius;
    h->decrease_factor = 2.0;
    h->invalid_run = 0;
    h->linearized = false; h->x_cost_known = false;
    h->scale_init = false;
    h->points_only_ready = false;
    h->points_resident_ready = false;
    h->campre_current = false;
    h->have_state = true;
    return SOSLAM_OK;
}

int soslam_ba_get_state(soslam_ba* h, double* poses, double* points)
{
    if (!h) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_state) { set_last_error("get_state before set_state"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    const size_t cam_bytes = sizeof(double) * 6 * h->n_cam, pt_bytes = sizeof(double) * 3 * (size_t)h->n_pt;
    if (cam_bytes + pt_bytes <= kPackedUploadMax && !getenv("SOSLAM_NO_PACKED_UPLOAD")) {
        // small states: one kernel gathers both arrays into pinned memory
        double* stage = nullptr;
        SOSLAM_CHECK(state_staging(h, cam_bytes + pt_bytes, &stage));
        PackedSeg* table = reinterpret_cast<PackedSeg*>(h->up_host[2]);
        table[0] = PackedSeg{stage, (uint64_t)reinterpret_cast<uintptr_t>(h->cams[h->cur].p), cam_bytes};
        table[1] = PackedSeg{stage + 6 * (size_t)h->n_cam, (uint64_t)reinterpret_cast<uintptr_t>(h->pts[h->cur].p), pt_bytes};
        launch_packed_scatter(h->stream, table, h->n_pt ? 2 : 1, nullptr);
        SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (poses) std::memcpy(poses, stage, cam_bytes);
        const double* sp = stage + 6 * (size_t)h->n_cam;
        if (points)
            for (uint32_t i = 0; i < h->n_pt; i++) std::memcpy(points + 3 * (size_t)h->pt_int2user[i], sp + 3 * (size_t)i, 3 * sizeof(double));
        return SOSLAM_OK;
    }
    std::vector<double> c((size_t)h->n_cam * 6), p((size_t)h->n_pt * 3);
    SOSLAM_HIP_CHECK(hipMemcpyAsync(c.data(), h->cams[h->cur].p, sizeof(double) * c.size(), hipMemcpyDeviceToHost, h->stream));
    if (h->n_pt) SOSLAM_HIP_CHECK(hipMemcpyAsync(p.data(), h->pts[h->cur].p, sizeof(double) * p.size(), hipMemcpyDeviceToHost, h->stream));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (poses) std::memcpy(poses, c.data(), sizeof(double) * c.size());
    if (points)
        for (uint32_t i = 0; i < h->n_pt; i++) std::memcpy(points + 3 * (size_t)h->pt_int2user[i], &p[3 * (size_t)i], 3 * sizeof(double));
    return SOSLAM_OK;
}

int soslam_ba_solve(soslam_ba* h, soslam_ba_summary* summary)
{
    if (!h) return SOSLAM_ERR_INVALID_ARGUMENT;
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    return run_lm(h, -1, summary);
}

int soslam_ba_iterate(soslam_ba* h, int32_t n, soslam_ba_summary* summary)
{
    if (!h || n < 0) return SOSLAM_ERR_INVALID_ARGUMENT;
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    return run_lm(h, n, summary);
}

int soslam_ba_get_iteration_log(soslam_ba* h, soslam_ba_iteration* out, int32_t capacity, int32_t* count)
{
    if (!h || !count) return SOSLAM_ERR_INVALID_ARGUMENT;
    *count = (int32_t)h->log.size();
    if (out) for (int32_t i = 0; i < capacity && i < *count; i++) out[i] = h->log[(size_t)i];
    return SOSLAM_OK;
}

int soslam_ba_optimize(const soslam_ba_options* opts, const double* pl, const double* pr, uint32_t n_cam, double* poses,
                       uint32_t n_pt, double* points, uint32_t n_obs, const uint32_t* obs_cam, const uint32_t* obs_pt,
                       const float* obs_uv, const uint8_t* cam_fixed, soslam_ba_summary* summary)
{
    soslam_ba* h = nullptr;
    SOSLAM_CHECK(soslam_ba_create(opts, &h));
    int st = soslam_ba_set_projection(h, pl, pr);
    if (st == SOSLAM_OK) st = soslam_ba_set_problem(h, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_uv, cam_fixed);
    if (st == SOSLAM_OK) st = soslam_ba_set_state(h, poses, points);
    if (st == SOSLAM_OK) st = soslam_ba_solve(h, summary);
    if (st == SOSLAM_OK) st = soslam_ba_get_state(h, poses, points);
    soslam_ba_destroy(h);
    return st;
}

int soslam_ba_set_allreduce(soslam_ba* h, soslam_allreduce_fn fn, void* user, int32_t rank, int32_t world)
{
    if (!h || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return SOSLAM_ERR_INVALID_ARGUMENT;
    h->allreduce = fn; h->allreduce_user = user; h->rank = rank; h->world = world;
    h->linearized = false; h->x_cost_known = false;   // sums over a different set of ranks from now on
    return SOSLAM_OK;
}

int soslam_ba_set_host_allreduce(soslam_ba* h, soslam_host_allreduce_fn fn, void* user, int32_t rank, int32_t world)
{
    if (!h || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) return SOSLAM_ERR_INVALID_ARGUMENT;
    h->host_allreduce = fn; h->host_allreduce_user = user; h->rank = rank; h->world = world;
    h->linearized = false; h->x_cost_known = false;
    return SOSLAM_OK;
}

int soslam_rccl_get_unique_id(void* id128)
{
    if (!id128) return SOSLAM_ERR_INVALID_ARGUMENT;
    return rccl_get_unique_id(id128);
}

int soslam_ba_init_rccl(soslam_ba* h, const void* id128, int32_t rank, int32_t world)
{
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (h->stream) SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    rccl_comm_destroy(h->rccl);
    h->rccl = nullptr;
    SOSLAM_CHECK(rccl_comm_create(id128, rank, world, h->device, &h->rccl));
    h->rank = rank; h->world = world;
    h->linearized = false; h->x_cost_known = false;   // sums over a different set of ranks from now on
    return SOSLAM_OK;
}

int soslam_ba_get_state_global(soslam_ba* h, double* poses, uint32_t n_pt_global, uint32_t shard_begin, double* points_global)
{
    if (!h || !points_global) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_state) { set_last_error("get_state_global before set_state"); return SOSLAM_ERR_STATE; }
    if ((uint64_t)shard_begin + h->n_pt > n_pt_global) {
        set_last_error("shard [%u, %u) does not fit %u points", shard_begin, shard_begin + h->n_pt, n_pt_global);
        return SOSLAM_ERR_INVALID_ARGUMENT;
    }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // own points in the caller's order at their global place, zero elsewhere; the sum over ranks is the whole map
    std::vector<double> own((size_t)h->n_pt * 3), p((size_t)h->n_pt * 3);
    if (h->n_pt) SOSLAM_HIP_CHECK(hipMemcpyAsync(p.data(), h->pts[h->cur].p, sizeof(double) * p.size(), hipMemcpyDeviceToHost, s));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < h->n_pt; i++) std::memcpy(&own[3 * (size_t)h->pt_int2user[i]], &p[3 * (size_t)i], 3 * sizeof(double));
    SOSLAM_CHECK(h->gather.alloc((size_t)n_pt_global * 3));
    SOSLAM_CHECK(h->gather.zero(s));
    if (h->n_pt) SOSLAM_HIP_CHECK(hipMemcpyAsync(h->gather.p + 3 * (size_t)shard_begin, own.data(), sizeof(double) * own.size(), hipMemcpyHostToDevice, s));
    SOSLAM_CHECK(do_allreduce(h, h->gather.p, (uint64_t)n_pt_global * 3, SOSLAM_REDUCE_SUM));
    SOSLAM_HIP_CHECK(hipMemcpyAsync(points_global, h->gather.p, sizeof(double) * 3 * (size_t)n_pt_global, hipMemcpyDeviceToHost, s));
    if (poses) SOSLAM_HIP_CHECK(hipMemcpyAsync(poses, h->cams[h->cur].p, sizeof(double) * 6 * h->n_cam, hipMemcpyDeviceToHost, s));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return SOSLAM_OK;
}

int soslam_ba_agree_status(soslam_ba* h, int local_status, int* agreed)
{
    if (!h || !agreed) return SOSLAM_ERR_INVALID_ARGUMENT;
    *agreed = local_status;
    if (!h->collective()) return SOSLAM_OK;
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    double* word = nullptr;
    if (h->rccl || h->host_allreduce) {
        SOSLAM_CHECK(h->agree.alloc(8));
        word = h->agree.p;
    } else if (h->have_problem && h->reduce) {
        word = h->scalp() + SC_STOP;    // a device callback may only accept ranges of the reduce buffer it was given
    } else {
        set_last_error("agree_status: the device-callback leg has no buffer before set_problem");
        return SOSLAM_ERR_COMM;
    }
    const double mine = local_status == SOSLAM_OK ? 0.0 : (double)local_status;
    double all = 0.0;
    SOSLAM_HIP_CHECK(hipMemcpyAsync(word, &mine, sizeof mine, hipMemcpyHostToDevice, s));
    SOSLAM_CHECK(do_allreduce(h, word, 1, SOSLAM_REDUCE_MAX));
    SOSLAM_HIP_CHECK(hipMemcpyAsync(&all, word, sizeof all, hipMemcpyDeviceToHost, s));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (word != h->agree.p) SOSLAM_HIP_CHECK(hipMemsetAsync(word, 0, sizeof(double), s));
    *agreed = (int)all;
    return SOSLAM_OK;
}

int soslam_ba_reduce_buffer_count(soslam_ba* h, uint64_t* count)
{
    if (!h || !count) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_problem) { set_last_error("reduce_buffer_count before set_problem"); return SOSLAM_ERR_STATE; }
    *count = h->reduce_count;
    return SOSLAM_OK;
}

int soslam_ba_set_reduce_buffer(soslam_ba* h, void* ptr, uint64_t count)
{
    if (!h) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_problem) { set_last_error("set_reduce_buffer before set_problem"); return SOSLAM_ERR_STATE; }
    if (!ptr) { h->reduce = h->reduce_own.p; h->linearized = false; h->x_cost_known = false; return SOSLAM_OK; }
    if (count < h->reduce_count) { set_last_error("reduce buffer too small: %llu < %llu f64", (unsigned long long)count, (unsigned long long)h->reduce_count); return SOSLAM_ERR_INVALID_ARGUMENT; }
    h->reduce = static_cast<double*>(ptr);
    SOSLAM_HIP_CHECK(hipMemsetAsync(h->reduce, 0, sizeof(double) * h->reduce_count, h->stream));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->linearized = false; h->x_cost_known = false;
    return SOSLAM_OK;
}

int soslam_ba_set_covisibility(soslam_ba* h, uint64_t n_pairs, const uint32_t* cam_a, const uint32_t* cam_b)
{
    if (!h || (n_pairs && (!cam_a || !cam_b))) return SOSLAM_ERR_INVALID_ARGUMENT;
    h->covis.clear();
    h->covis.reserve(n_pairs);
    for (uint64_t i = 0; i < n_pairs; i++) h->covis.emplace_back(cam_a[i], cam_b[i]);
    return SOSLAM_OK;
}

void soslam_ba_shard_range(uint32_t n_pt, int32_t rank, int32_t world, uint32_t* begin, uint32_t* end)
{
    if (world < 1) world = 1;
    if (begin) *begin = (uint32_t)(((uint64_t)rank * n_pt) / (uint64_t)world);
    if (end) *end = (uint32_t)(((uint64_t)(rank + 1) * n_pt) / (uint64_t)world);
}

int soslam_ba_time_kernel(soslam_ba* h, int32_t kernel, int32_t reps, float* avg_ms)
{
    if (!h || reps < 1 || !avg_ms) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_problem || !h->have_state || !h->have_proj) { set_last_error("time_kernel needs a problem and a state"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (!h->linearized) SOSLAM_CHECK(linearize(h));
    if (kernel == SOSLAM_KERNEL_SCHUR || kernel == SOSLAM_KERNEL_BACKSUB) {
        // these read the point scales; make sure they exist
        if (!h->scale_init) launch_point_scale(s, h->n_pt, h->C.p, h->opt.jacobi_scaling, h->sp.p);
    }
    const LmDiag lm = lm_diag(h, h->radius > 0 ? h->radius : h->opt.initial_radius);
    hipEvent_t e0, e1;
    SOSLAM_HIP_CHECK(hipEventCreate(&e0));
    SOSLAM_HIP_CHECK(hipEventCreate(&e1));
    auto once = [&]() {
        switch (kernel) {
        case SOSLAM_KERNEL_LINEARIZE:
            launch_linearize(s, h->n_tiles, h->tiles.p, h->uv.p, h->obs_pt.p, h->campre.p, h->pts[h->cur].p, h->cam_free.p,
                             h->proj, h->opt.huber_delta, h->rows(), h->tile_part.p);
            break;
        case SOSLAM_KERNEL_COST:
            launch_cost(s, h->n_tiles, h->tiles.p, h->uv.p, h->obs_pt.p, h->campre.p, h->pts[h->cur].p, h->proj,
                        h->opt.huber_delta, h->cost_part.p);
            break;
        case SOSLAM_KERNEL_POINT_REDUCE:
            launch_point_reduce(s, h->n_pt, h->pt_start.p, h->pt_obs.p, h->q_cam.p, h->rows(), h->campre.p, h->C.p, h->gp.p);
            break;
        case SOSLAM_KERNEL_SCHUR:
            launch_schur(s, h->kmax, h->n_chunks, h->chunks.p, h->batches.p, h->chunk_slab.p, h->chunk_cam.p, h->pair_row.p, h->pt_obs.p, h->q_pt.p,
                         h->q_slot.p, h->rows(), h->campre.p, h->pts[h->cur].p, h->C.p, h->gp.p, h->sp.p, lm, h->Cinv.p, h->ptfac.p, h->slab.p,
                         h->scalp(), h->pt_start.p, h->q_cam.p, points_fused(h) ? 1 : 0);
            break;
        case SOSLAM_KERNEL_BACKSUB:
            launch_backsub(s, h->n_pt, h->pt_start.p, h->pt_obs.p, h->q_cam.p, h->ar.p, h->campre.p, h->dcw.p, h->Cinv.p,
                           h->C.p, h->gp.p, h->sp.p, h->pts[h->cur].p, lm, h->opt.lower_bound, h->opt.upper_bound,
                           h->pts[h->cur ^ 1].p, h->dp.p, h->part.p);
            break;
        default: break;
        }
    };
    if (kernel < 0 || kernel > SOSLAM_KERNEL_BACKSUB) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return SOSLAM_ERR_INVALID_ARGUMENT; }
    if (kernel == SOSLAM_KERNEL_BACKSUB) {
        SOSLAM_HIP_CHECK(hipMemsetAsync(h->dcw.p, 0, sizeof(double) * 6 * h->n_cam, s));
        run_schur(h, lm);
    }
    once();  // warm-up
    SOSLAM_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) once();
    SOSLAM_HIP_CHECK(hipEventRecord(e1, s));
    SOSLAM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    SOSLAM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)reps;
    // the timed launches may have disturbed the reduce buffer / candidate; force a clean re-linearisation
    h->linearized = false; h->x_cost_known = false;
    SOSLAM_HIP_CHECK(hipGetLastError());
    return SOSLAM_OK;
}

int soslam_ba_debug_step(soslam_ba* h, double radius)
{
    if (!h || !(radius > 0.0)) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_problem || !h->have_state || !h->have_proj) { set_last_error("debug_step needs a problem and a state"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    h->scale_init = false;  // scaling from THIS linearisation, as on the first iteration
    h->x_cost_known = false;   // the debug read-back reports the cost the device summed
    SOSLAM_HIP_CHECK(hipMemsetAsync(h->scalp(), 0, sizeof(double) * SC_COUNT, h->stream));
    SOSLAM_CHECK(linearize(h));
    SOSLAM_CHECK(take_step(h, radius));
    return SOSLAM_OK;
}

int soslam_ba_debug_read(soslam_ba* h, int32_t what, void* dst, uint64_t bytes)
{
    if (!h || !dst) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_problem) { set_last_error("debug_read before set_problem"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    auto need = [&](uint64_t n) -> int {
        if (bytes < n) { set_last_error("debug_read: buffer of %llu bytes, %llu needed", (unsigned long long)bytes, (unsigned long long)n); return SOSLAM_ERR_INVALID_ARGUMENT; }
        return SOSLAM_OK;
    };
    double* out = static_cast<double*>(dst);
    const size_t n_obs = h->n_obs;
    switch (what) {
    case SOSLAM_DBG_RESIDUALS:
    case SOSLAM_DBG_JAC_POINT:
    case SOSLAM_DBG_JAC_CAM: {
        // the device keeps compact rows [G | h]; the blocks the tests compare are formed again by a read-back kernel
        // from ba_linearize's own per-observation function (residual_ad), at the linearisation point (pose table and
        // points it was made from); the stored rows are checked against them through SOSLAM_DBG_COMPACT_ROWS
        const size_t w = what == SOSLAM_DBG_RESIDUALS ? 4 : (what == SOSLAM_DBG_JAC_POINT ? 12 : 24);
        SOSLAM_CHECK(need(n_obs * w * sizeof(double)));
        DevBuf<double> dr, djc, djp;
        SOSLAM_CHECK(dr.alloc(n_obs * 4)); SOSLAM_CHECK(djc.alloc(n_obs * 24)); SOSLAM_CHECK(djp.alloc(n_obs * 12));
        launch_debug_rows(s, h->n_tiles, h->tiles.p, h->uv.p, h->obs_pt.p, h->campre.p, h->pts[h->cur].p, h->cam_free.p, h->proj,
                          h->opt.huber_delta, dr.p, djc.p, djp.p);
        std::vector<double> tmp(n_obs * w);
        const double* src = what == SOSLAM_DBG_RESIDUALS ? dr.p : (what == SOSLAM_DBG_JAC_POINT ? djp.p : djc.p);
        SOSLAM_HIP_CHECK(hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < n_obs; i++) std::memcpy(out + w * h->obs_int2user[i], &tmp[i * w], w * sizeof(double));
        return SOSLAM_OK;
    }
    case SOSLAM_DBG_COMPACT_ROWS: {
        // the stored rows themselves (whatever the production linearisation left in `ar`), not a recomputation
        SOSLAM_CHECK(need(n_obs * 9 * sizeof(double)));
        if (!h->linearized) { set_last_error("debug_read(COMPACT_ROWS): no linearisation held"); return SOSLAM_ERR_STATE; }
        std::vector<double> tmp(n_obs * kArRow);
        SOSLAM_HIP_CHECK(hipMemcpyAsync(tmp.data(), h->ar.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < n_obs; i++) {   // [G (6) | h (3)] per observation from the two arrays (ba_kernels.h: CompactRows)
            std::memcpy(out + 9 * (size_t)h->obs_int2user[i], &tmp[i * kArG], 6 * sizeof(double));
            std::memcpy(out + 9 * (size_t)h->obs_int2user[i] + 6, &tmp[n_obs * kArG + i * kArH], 3 * sizeof(double));
        }
        return SOSLAM_OK;
    }
    case SOSLAM_DBG_COST:
        SOSLAM_CHECK(need(sizeof(double)));
        SOSLAM_HIP_CHECK(hipMemcpyAsync(out, h->scalp() + SC_COST_X, sizeof(double), hipMemcpyDeviceToHost, s));
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        return SOSLAM_OK;
    case SOSLAM_DBG_S_DENSE: {
        const size_t n6 = (size_t)h->n_free * 6;
        SOSLAM_CHECK(need(n6 * n6 * sizeof(double)));
        std::vector<double> blk((size_t)h->n_blocks * 36);
        SOSLAM_HIP_CHECK(hipMemcpyAsync(blk.data(), h->S(), blk.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        std::memset(out, 0, n6 * n6 * sizeof(double));
        for (uint32_t b = 0; b < h->n_blocks; b++)
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 6; c++) {
                    const size_t i = 6 * (size_t)h->h_blk_row[b] + a, j = 6 * (size_t)h->h_blk_col[b] + c;
                    out[i * n6 + j] = blk[36 * (size_t)b + a * 6 + c];
                    out[j * n6 + i] = blk[36 * (size_t)b + a * 6 + c];
                }
        return SOSLAM_OK;
    }
    case SOSLAM_DBG_RHS:
        SOSLAM_CHECK(need((size_t)h->n_free * 6 * sizeof(double)));
        SOSLAM_HIP_CHECK(hipMemcpyAsync(out, h->rhs(), (size_t)h->n_free * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        return SOSLAM_OK;
    case SOSLAM_DBG_STEP_CAM:
        SOSLAM_CHECK(need((size_t)h->n_cam * 6 * sizeof(double)));
        SOSLAM_HIP_CHECK(hipMemcpyAsync(out, h->dc_full.p, (size_t)h->n_cam * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        return SOSLAM_OK;
    case SOSLAM_DBG_STEP_POINT: {
        SOSLAM_CHECK(need((size_t)h->n_pt * 3 * sizeof(double)));
        std::vector<double> tmp((size_t)h->n_pt * 3);
        SOSLAM_HIP_CHECK(hipMemcpyAsync(tmp.data(), h->dp.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < h->n_pt; i++) std::memcpy(out + 3 * (size_t)h->pt_int2user[i], &tmp[3 * (size_t)i], 3 * sizeof(double));
        return SOSLAM_OK;
    }
    case SOSLAM_DBG_STEP_SCALARS: {
        SOSLAM_CHECK(need(6 * sizeof(double)));
        const StepScalars sc = read_scalars(h);
        out[0] = sc.x_cost; out[1] = sc.mcc; out[2] = sc.cand_cost; out[3] = sc.step_norm;
        out[4] = (double)sc.lin_iters; out[5] = (double)sc.lin_status;
        return SOSLAM_OK;
    }
    default:
        return SOSLAM_ERR_INVALID_ARGUMENT;
    }
}

void soslam_pose_from_global_matrix(const float* t_wc16, double* pose6)
{
    soslam_host::Mat4f m;
    std::memcpy(m.m, t_wc16, sizeof m.m);
    std::array<double, 6> p;
    soslam_host::MatrixToPose(m.inverse(), p);
    std::memcpy(pose6, p.data(), sizeof(double) * 6);
}

void soslam_global_matrix_from_pose(const double* pose6, float* t_wc16)
{
    std::array<double, 6> p;
    std::memcpy(p.data(), pose6, sizeof(double) * 6);
    soslam_host::Mat4f m;
    soslam_host::PoseToMatrix(p, m);
    const soslam_host::Mat4f inv = m.inverse();
    std::memcpy(t_wc16, inv.m, sizeof inv.m);
}

}  // extern "C"
