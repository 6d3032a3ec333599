// pg_solver.hip - host side of the pose-graph backend: block pattern of H from the edge list, device state, g2o's
// Levenberg controller (OptimizationAlgorithmLevenberg::solve) and the C ABI of
// include/soslam_pg.h.  Replaces m_optimizer.initializeOptimization(); m_optimizer.optimize(10)
// (/root/reference/src/pose_graph_optimizer.cpp:68-69).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>

#include "common.h"
#include "ba_kernels.h"   // the SC_* slots the band solver reports through
#include "linsolve.h"
#include "pg_kernels.h"
#include "soslam_pg.h"

using namespace soslam;

struct soslam_pg {
    soslam_pg_options opt{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int device = 0;
    bool have_graph = false;
    uint32_t n_vertex = 0, n_edge = 0, n_free = 0, n_blocks = 0, n_chi_part = 0, n_scale_part = 0;
    PgInfo info{};
    std::vector<int32_t> h_free;
    std::vector<uint32_t> h_blk_row, h_blk_col;
    // host mirror of the graph: soslam_pg_append extends it (the reference's g2o optimizer persists and grows between
    // calls, /root/reference/src/pose_graph_optimizer.cpp:56-59); estimates of existing vertices live on the device
    std::vector<double> g_est, g_meas;
    std::vector<uint8_t> g_fixed;
    std::vector<uint32_t> g_ef, g_et;
    DevBuf<double> est[2], meas, H, b, x, resid, work, chi_part, chi_part_lin, scale_part, scal, econ;
    DevBuf<uint32_t> ef, et, row_ptr, ent_col, ent_blk, g_ptr, g_ent, blk_row, blk_col;
    DevBuf<uint8_t> ent_trans;
    DevBuf<int32_t> free_idx, diag_block;
    DevBuf<PgEdgeBlocks> eb;
    // two-level preconditioner (pcg2_solve): aggregates of neighbouring free vertices, six rigid-body modes each
    bool two_level = false;
    int last_lin_it = 0;                // PCG iterations of the last solve: the next solve enqueues that many before it looks
    uint32_t n_agg = 0, ncp = 0, n_cb = 0;
    DevBuf<uint32_t> agg_ptr, row_agg, agg_ref, free_vertex, cb_ptr, cb_ent, cb_I, cb_J;
    DevBuf<double> cP, cG, cAc0, cAinv[2], cebuf[2], crc, cstatus;
    // the coarse inverse of a Levenberg trial is computed on a second stream while that trial's PCG runs with the previous one
    // (any SPD coarse operator preconditions): cinv_cur holds the newest complete one
    hipStream_t cstream = nullptr;
    hipEvent_t ev_coarse_in = nullptr, ev_coarse_done = nullptr;
    int cinv_cur = 0;
    bool cinv_valid = false, cinv_pending = false;
    // chain-shaped graphs (the reference's: keyframes in order, a few loop closures): the band of the odometry chain factored
    // exactly by block cyclic reduction, the closure blocks left to the PCG's matrix-vector product (the BA path's treatment of
    // loop closures).  The band part of H is positive definite by itself: dropping a closure edge's OFF-diagonal
    // block leaves its two diagonal contributions, which are positive semidefinite
    bool band_mode = false, band_off = false;
    int band_bw = 0, band_rounds = 2;
    double applied_shift = 0.0;         // the Levenberg shift currently inside H's diagonal (band mode)
    DevBuf<int32_t> cr_map;
    DevBuf<double> cr_ws, lin_scal;
    int cur = 0;
    double setup_seconds = 0.0;
    std::vector<soslam_pg_iteration> log;
    hipEvent_t ev[2] = {nullptr, nullptr};

    ~soslam_pg()
    {
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
        if (cstream) { (void)hipStreamSynchronize(cstream); (void)hipStreamDestroy(cstream); }
        if (ev_coarse_in) (void)hipEventDestroy(ev_coarse_in);
        if (ev_coarse_done) (void)hipEventDestroy(ev_coarse_done);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

double now_sec()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

BsrView bsr_view(const soslam_pg* h)
{
    return BsrView{h->n_free, h->row_ptr.p, h->ent_col.p, h->ent_blk.p, h->ent_trans.p, h->diag_block.p, h->H.p};
}

int build_graph(soslam_pg* h, uint32_t n_vertex, const double* est, const uint8_t* fixed, uint32_t n_edge, const uint32_t* ef,
                const uint32_t* et, const double* meas, const double* info)
{
    const double t0 = now_sec();
    hipStream_t s = h->stream;
    for (uint32_t k = 0; k < n_edge; k++)
        if (ef[k] >= n_vertex || et[k] >= n_vertex || ef[k] == et[k]) {
            set_last_error("edge %u (%u -> %u) is out of range or a self loop", k, ef[k], et[k]);
            return SOSLAM_ERR_INVALID_ARGUMENT;
        }
    h->n_vertex = n_vertex; h->n_edge = n_edge;
    std::memcpy(h->info.m, info, sizeof h->info.m);
    h->h_free.assign(n_vertex, -1);
    uint32_t nf = 0;
    for (uint32_t v = 0; v < n_vertex; v++)
        if (!(fixed && fixed[v])) h->h_free[v] = (int32_t)nf++;
    h->n_free = nf;
    // Aggregates for the two-level preconditioner: greedy breadth-first growth over the graph of free vertices, at most
    // kAggMax vertices each (one workgroup of pcg2_solve per aggregate), compact in the graph whatever the vertex numbering
    // (a lawn-mower path, a chain with loop closures).  The free indices are then renumbered aggregate by aggregate, so that
    // an aggregate is a contiguous range of block rows of H.
    constexpr uint32_t kAggMax = 42;   // block rows of one pcg2 workgroup
    const int pre = h->opt.preconditioner;
    // A chain with loop closures?  Ordered breadth-first (Cuthill-McKee: from the first free vertex, neighbours by ascending degree) a
    // chain of keyframes with a handful of loop edges is a BAND whatever the length of the loops - a loop's two arms are numbered
    // alternately, the closure edge joins neighbours - so H has an exact band factor and a solve is ONE round.  Edges the band of ten
    // still leaves out (loops nested several deep) stay in the matrix-vector product, about a dozen PCG rounds apiece.
    h->band_mode = false; h->band_off = false; h->band_bw = 0;
    // (the factor's workspace is 187 KB per nine vertices: 4 GB at 200 000 vertices, where this path ends)
    if (nf >= 32 && nf <= 200000 && (pre == SOSLAM_PG_PRECOND_AUTO || pre == SOSLAM_PG_PRECOND_BAND_FACTOR) && std::getenv("SOSLAM_PG_NO_BAND") == nullptr) {
        std::vector<std::vector<uint32_t>> adj(n_vertex);
        for (uint32_t k = 0; k < n_edge; k++)
            if (h->h_free[ef[k]] >= 0 && h->h_free[et[k]] >= 0) { adj[ef[k]].push_back(et[k]); adj[et[k]].push_back(ef[k]); }
        std::vector<int32_t> pos(n_vertex, -1);
        {
            std::vector<uint32_t> queue;
            queue.reserve(nf);
            for (uint32_t v0 = 0; v0 < n_vertex; v0++) {
                if (h->h_free[v0] < 0 || pos[v0] >= 0) continue;
                pos[v0] = (int32_t)queue.size();
                queue.push_back(v0);
                for (size_t qi = (size_t)pos[v0]; qi < queue.size(); qi++) {
                    const uint32_t v = queue[qi];
                    std::vector<uint32_t>& nb = adj[v];
                    std::sort(nb.begin(), nb.end(), [&](uint32_t x, uint32_t y) { return adj[x].size() != adj[y].size() ? adj[x].size() < adj[y].size() : x < y; });
                    for (uint32_t u : nb)
                        if (pos[u] < 0) { pos[u] = (int32_t)queue.size(); queue.push_back(u); }
                }
            }
        }
        // the better of the two orders: breadth-first, or as given (the reference numbers keyframes as they arrive)
        auto outside_of = [&](const std::vector<int32_t>& idx, uint64_t* pairs_out) {
            uint64_t pairs = 0, in = 0;
            for (uint32_t k = 0; k < n_edge; k++) {
                const int32_t a_ = idx[ef[k]], b_ = idx[et[k]];
                if (a_ < 0 || b_ < 0) continue;
                pairs++;
                if (std::abs(a_ - b_) <= kCrBandMax) in++;
            }
            *pairs_out = pairs;
            return pairs - in;
        };
        uint64_t pairs = 0, pairs2 = 0;
        const uint64_t out_bfs = outside_of(pos, &pairs), out_nat = outside_of(h->h_free, &pairs2);
        const bool use_bfs = out_bfs < out_nat && std::getenv("SOSLAM_PG_NO_BFS") == nullptr;   // (the switch: tests of the off-band path)
        const std::vector<int32_t>& idx = use_bfs ? pos : h->h_free;
        const uint64_t outside = use_bfs ? out_bfs : out_nat;
        // Every closure left to the matrix-vector product costs the PCG about a dozen rounds (it perturbs the preconditioned matrix by
        // rank 12, and not by little), a round 95 us at 2 000 vertices against 18 us for an iteration of the two-level PCG, which needs
        // ~270 of them on such a chain whatever the closures: measured in the order given (scripts/pg_chain_probe.py, ten iterations)
        // 14 / 23 / 31 / 35 / 48 / 68 ms at 1 / 2 / 3 / 4 / 6 / 12 closures against 44-49 ms.  AUTO takes the band with up to four edges outside it
        const bool fits = pairs > 0 && outside * 100 <= pairs && (pre == SOSLAM_PG_PRECOND_BAND_FACTOR || outside <= 4);
        if (fits || (pre == SOSLAM_PG_PRECOND_BAND_FACTOR && pairs > 0)) {
            h->band_mode = true;
            int wmax = 1;
            for (uint32_t k = 0; k < n_edge; k++) {
                const int32_t a_ = idx[ef[k]], b_ = idx[et[k]];
                if (a_ >= 0 && b_ >= 0 && std::abs(a_ - b_) <= kCrBandMax) wmax = std::max(wmax, std::abs(a_ - b_));
            }
            // super-blocks of nine vertices whatever the band's own width (a tridiagonal chain factored in 6 x 6 nodes would be a tree
            // of log2(n) levels of tiny inverses): ten only when the band needs it
            h->band_bw = wmax <= 9 ? 9 : kCrBandMax;
            uint64_t in = 0;
            for (uint32_t k = 0; k < n_edge; k++) {
                const int32_t a_ = idx[ef[k]], b_ = idx[et[k]];
                if (a_ >= 0 && b_ >= 0 && std::abs(a_ - b_) <= h->band_bw) in++;
            }
            h->band_off = in < pairs;
            h->band_rounds = 2;
            if (use_bfs) h->h_free = pos;   // the free index IS the position in H
        }
    }
    h->two_level = !h->band_mode && nf > 0 && (pre == SOSLAM_PG_PRECOND_TWO_LEVEL || (pre == SOSLAM_PG_PRECOND_AUTO && nf >= 64)) && (nf + kAggMax - 1) / kAggMax * 6 <= 1200;
    std::vector<uint32_t> agg_ptr, row_agg, agg_ref, free_vertex(nf);
    if (h->two_level) {
        std::vector<std::vector<uint32_t>> adj(n_vertex);
        for (uint32_t k = 0; k < n_edge; k++)
            if (h->h_free[ef[k]] >= 0 && h->h_free[et[k]] >= 0) { adj[ef[k]].push_back(et[k]); adj[et[k]].push_back(ef[k]); }
        // aggregate sizes: as even as the cap allows (a count of ceil(nf / kAggMax) aggregates, sizes within one of each other
        // would need a partitioner; breadth-first growth to the target size, leftovers joined to a neighbouring aggregate or kept)
        // growth stops here; what is left between grown aggregates is merged into them up to kAggMax.  configs[4], ten iterations
        // (scripts/pg_agg_probe.py): targets 28 / 30 / 32 / 34 / 38 / 40 / 42 give 29.0 / 28.2 / 28.7 / 30.6 / 28.4 / 27.7 / 31.5 ms - a noisy
        // function of where the cuts fall, flat between 28 and 40
        uint32_t target = 30;
        if (const char* e = std::getenv("SOSLAM_PG_AGG")) target = (uint32_t)std::min(std::max(std::atoi(e), 4), (int)kAggMax);   // development
        std::vector<int32_t> agg_of(n_vertex, -1);
        std::vector<std::vector<uint32_t>> members;
        std::vector<uint32_t> queue;
        for (uint32_t seed = 0; seed < n_vertex; seed++) {
            if (h->h_free[seed] < 0 || agg_of[seed] >= 0) continue;
            const int32_t id = (int32_t)members.size();
            members.emplace_back();
            queue.clear();
            queue.push_back(seed);
            agg_of[seed] = id;
            for (size_t qi = 0; qi < queue.size() && members.back().size() < target; qi++) {
                const uint32_t v = queue[qi];
                members.back().push_back(v);
                for (uint32_t u : adj[v])
                    if (agg_of[u] < 0 && queue.size() < target) { agg_of[u] = id; queue.push_back(u); }
            }
            // vertices claimed for the queue but not reached (the size cap came first) are released
            for (size_t qi = members.back().size(); qi < queue.size(); qi++) agg_of[queue[qi]] = -1;
            for (uint32_t v : members.back()) agg_of[v] = id;
        }
        // leftovers (pieces enclosed by grown aggregates) and small aggregates join the smallest neighbouring aggregate that has
        // room, smallest pieces first, until nothing fits any more
        for (bool merged = true; merged;) {
            merged = false;
            std::vector<size_t> order;
            for (size_t a = 0; a < members.size(); a++)
                if (!members[a].empty() && members[a].size() < target) order.push_back(a);
            std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return members[x].size() != members[y].size() ? members[x].size() < members[y].size() : x < y; });
            for (size_t a : order) {
                if (members[a].empty()) continue;
                int32_t best = -1;
                for (uint32_t v : members[a])
                    for (uint32_t u : adj[v]) {
                        const int32_t b = agg_of[u];
                        if (b >= 0 && b != (int32_t)a && !members[(size_t)b].empty() && members[(size_t)b].size() + members[a].size() <= kAggMax &&
                            (best < 0 || members[(size_t)b].size() < members[(size_t)best].size() ||
                             (members[(size_t)b].size() == members[(size_t)best].size() && b < best))) best = b;
                    }
                if (best >= 0) {
                    for (uint32_t v : members[a]) { agg_of[v] = best; members[(size_t)best].push_back(v); }
                    members[a].clear();
                    merged = true;
                }
            }
        }
        uint32_t next = 0;
        agg_ptr.push_back(0);
        for (auto& m : members) {
            if (m.empty()) continue;
            std::sort(m.begin(), m.end());
            agg_ref.push_back(m.front());
            for (uint32_t v : m) { h->h_free[v] = (int32_t)next; free_vertex[next] = v; row_agg.push_back((uint32_t)agg_ptr.size() - 1); next++; }
            agg_ptr.push_back(next);
        }
        h->n_agg = (uint32_t)agg_ptr.size() - 1;
        h->last_lin_it = 0;
        h->ncp = (h->n_agg * 6 + 59) / 60 * 60;
        // the coarse kernels hold P^T r of every aggregate in LDS (1 260 entries): a graph that breaks into more pieces than that
        // (many small components) keeps the aggregate numbering and takes block-Jacobi
        if (h->ncp > 1260) { h->two_level = false; h->n_agg = 0; h->ncp = 0; }
    } else {
        h->n_agg = 0; h->ncp = 0;
    }

    // upper block pattern: diagonal + one block per connected pair of free vertices
    std::vector<std::vector<uint32_t>> rows(nf);
    for (uint32_t f = 0; f < nf; f++) rows[f].push_back(f);
    for (uint32_t k = 0; k < n_edge; k++) {
        const int32_t a = h->h_free[ef[k]], b = h->h_free[et[k]];
        if (a >= 0 && b >= 0) rows[std::min(a, b)].push_back((uint32_t)std::max(a, b));
    }
    std::vector<uint32_t> row_first(nf + 1, 0);
    for (uint32_t f = 0; f < nf; f++) {
        std::sort(rows[f].begin(), rows[f].end());
        rows[f].erase(std::unique(rows[f].begin(), rows[f].end()), rows[f].end());
        row_first[f + 1] = row_first[f] + (uint32_t)rows[f].size();
    }
    h->n_blocks = row_first[nf];
    h->h_blk_row.resize(h->n_blocks); h->h_blk_col.resize(h->n_blocks);
    std::vector<int32_t> diag_block(nf);
    for (uint32_t f = 0; f < nf; f++)
        for (size_t e = 0; e < rows[f].size(); e++) {
            const uint32_t blk = row_first[f] + (uint32_t)e;
            h->h_blk_row[blk] = f; h->h_blk_col[blk] = rows[f][e];
            if (rows[f][e] == f) diag_block[f] = (int32_t)blk;
        }
    auto find_block = [&](uint32_t i, uint32_t j) -> int32_t {
        const auto& r = rows[i];
        auto it = std::lower_bound(r.begin(), r.end(), j);
        return (int32_t)(row_first[i] + (uint32_t)(it - r.begin()));
    };
    std::vector<uint32_t> cb_ptr, cb_ent, cb_I, cb_J;
    if (h->two_level) {
        two_level_lists(h->n_agg, row_agg.data(), h->n_blocks, h->h_blk_row.data(), h->h_blk_col.data(), cb_ptr, cb_ent, cb_I, cb_J);
        h->n_cb = (uint32_t)cb_I.size();
    }
    std::vector<PgEdgeBlocks> eb(n_edge);
    for (uint32_t k = 0; k < n_edge; k++) {
        const int32_t a = h->h_free[ef[k]], b = h->h_free[et[k]];
        PgEdgeBlocks e{a, b, a >= 0 ? diag_block[a] : -1, b >= 0 ? diag_block[b] : -1, -1, 0};
        if (a >= 0 && b >= 0) {
            e.off = find_block((uint32_t)std::min(a, b), (uint32_t)std::max(a, b));
            e.off_is_ji = b < a ? 1 : 0;
        }
        eb[k] = e;
    }
    // gather lists of pg_gather: per block of H the edge records that feed it, in edge order
    std::vector<uint32_t> g_ptr(h->n_blocks + 1, 0), g_ent;
    {
        std::vector<std::vector<uint32_t>> lists(h->n_blocks);
        for (uint32_t k = 0; k < n_edge; k++) {
            const PgEdgeBlocks& e = eb[k];
            if (e.fi >= 0) lists[(size_t)e.diag_i].push_back(2 * k);
            if (e.fj >= 0) lists[(size_t)e.diag_j].push_back(2 * k + 1);
            if (e.off >= 0) lists[(size_t)e.off].push_back(k);
        }
        for (uint32_t blk = 0; blk < h->n_blocks; blk++) {
            g_ptr[blk + 1] = g_ptr[blk] + (uint32_t)lists[blk].size();
            g_ent.insert(g_ent.end(), lists[blk].begin(), lists[blk].end());
        }
    }
    std::vector<uint32_t> row_ptr(nf + 1, 0), ent_col, ent_blk;
    std::vector<uint8_t> ent_trans;
    {
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> full(nf);
        for (uint32_t blk = 0; blk < h->n_blocks; blk++) {
            const uint32_t i = h->h_blk_row[blk], j = h->h_blk_col[blk];
            full[i].push_back({j, blk});
            if (i != j) full[j].push_back({i, blk | 0x80000000u});
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
    h->n_chi_part = div_up(n_edge, 128);
    h->n_scale_part = div_up(n_vertex, 256);
    std::vector<double> v_est(est, est + 7 * (size_t)n_vertex), v_meas(meas, meas + 7 * (size_t)n_edge);
    std::vector<uint32_t> v_ef(ef, ef + n_edge), v_et(et, et + n_edge);
    SOSLAM_CHECK(h->est[0].upload(v_est, s));
    SOSLAM_CHECK(h->est[1].alloc(v_est.size()));
    SOSLAM_CHECK(h->meas.upload(v_meas, s));
    SOSLAM_CHECK(h->ef.upload(v_ef, s));
    SOSLAM_CHECK(h->et.upload(v_et, s));
    SOSLAM_CHECK(h->free_idx.upload(h->h_free, s));
    SOSLAM_CHECK(h->diag_block.upload(diag_block, s));
    SOSLAM_CHECK(h->eb.upload(eb, s));
    SOSLAM_CHECK(h->row_ptr.upload(row_ptr, s));
    SOSLAM_CHECK(h->ent_col.upload(ent_col, s));
    SOSLAM_CHECK(h->ent_blk.upload(ent_blk, s));
    SOSLAM_CHECK(h->ent_trans.upload(ent_trans, s));
    SOSLAM_CHECK(h->g_ptr.upload(g_ptr, s));
    SOSLAM_CHECK(h->g_ent.upload(g_ent, s));
    SOSLAM_CHECK(h->blk_row.upload(h->h_blk_row, s));
    SOSLAM_CHECK(h->blk_col.upload(h->h_blk_col, s));
    SOSLAM_CHECK(h->econ.alloc((size_t)n_edge * kPgEdgeRec));
    SOSLAM_CHECK(h->H.alloc((size_t)h->n_blocks * 36));
    SOSLAM_CHECK(h->b.alloc((size_t)nf * 6));
    SOSLAM_CHECK(h->x.alloc((size_t)nf * 6));
    SOSLAM_CHECK(h->x.zero(s));
    SOSLAM_CHECK(h->resid.alloc((size_t)nf * 6));
    SOSLAM_CHECK(h->work.alloc(std::max({pcg_multi_work_count(nf), pcg2_work_count(nf, h->n_agg), h->band_mode ? pcg_band_work_count(nf) : (size_t)0})));
    if (h->band_mode) {
        std::vector<int32_t> map(cr_map_count(nf, h->band_bw));
        cr_build_map(nf, h->band_bw, h->n_blocks, h->h_blk_row.data(), h->h_blk_col.data(), map.data());
        SOSLAM_CHECK(h->cr_map.upload(map, s));
        SOSLAM_CHECK(h->cr_ws.alloc(cr_count(nf, h->band_bw)));
        SOSLAM_CHECK(h->lin_scal.alloc(SC_COUNT + 8));
        SOSLAM_CHECK(h->lin_scal.zero(s));
    }
    if (h->two_level) {
        SOSLAM_CHECK(h->agg_ptr.upload(agg_ptr, s));
        SOSLAM_CHECK(h->row_agg.upload(row_agg, s));
        SOSLAM_CHECK(h->agg_ref.upload(agg_ref, s));
        SOSLAM_CHECK(h->free_vertex.upload(free_vertex, s));
        SOSLAM_CHECK(h->cb_ptr.upload(cb_ptr, s));
        SOSLAM_CHECK(h->cb_ent.upload(cb_ent, s));
        SOSLAM_CHECK(h->cb_I.upload(cb_I, s));
        SOSLAM_CHECK(h->cb_J.upload(cb_J, s));
        SOSLAM_CHECK(h->cP.alloc((size_t)nf * 36));
        SOSLAM_CHECK(h->cG.alloc((size_t)h->n_agg * 36));
        SOSLAM_CHECK(h->cAc0.alloc((size_t)h->ncp * h->ncp));
        if (h->cstream) SOSLAM_HIP_CHECK(hipStreamSynchronize(h->cstream));
        else {
            SOSLAM_HIP_CHECK(hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
            SOSLAM_HIP_CHECK(hipEventCreateWithFlags(&h->ev_coarse_in, hipEventDisableTiming));
            SOSLAM_HIP_CHECK(hipEventCreateWithFlags(&h->ev_coarse_done, hipEventDisableTiming));
        }
        h->cinv_valid = false; h->cinv_pending = false; h->cinv_cur = 0;
        for (int q = 0; q < 2; q++) {
            SOSLAM_CHECK(h->cAinv[q].alloc((size_t)h->ncp * h->ncp));
            SOSLAM_CHECK(h->cebuf[q].alloc(2 * 3600));
        }
        SOSLAM_CHECK(h->cstatus.alloc(8));
        SOSLAM_CHECK(h->cstatus.zero(s));
        SOSLAM_CHECK(h->crc.alloc(h->ncp));
        SOSLAM_CHECK(h->crc.zero(s));
    }
    SOSLAM_CHECK(h->chi_part.alloc(std::max<uint32_t>(h->n_chi_part, div_up(n_edge, 256))));
    SOSLAM_CHECK(h->chi_part_lin.alloc(h->n_chi_part));
    SOSLAM_CHECK(h->scale_part.alloc(h->n_scale_part));
    SOSLAM_CHECK(h->scal.alloc(8));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
    h->cur = 0;
    h->have_graph = true;
    h->setup_seconds = now_sec() - t0;
    // keep the graph for soslam_pg_append (the estimates are refreshed from the device when it is called)
    h->g_est.assign(est, est + 7 * (size_t)n_vertex);
    h->g_meas.assign(meas, meas + 7 * (size_t)n_edge);
    h->g_fixed.assign(n_vertex, 0);
    if (fixed) h->g_fixed.assign(fixed, fixed + n_vertex);
    h->g_ef.assign(ef, ef + n_edge);
    h->g_et.assign(et, et + n_edge);
    return SOSLAM_OK;
}

// H, b and the robust chi2 at the current estimates; scal[0] = chi2, scal[1] = max |diag H|
int linearize(soslam_pg* h, double* dbg_e, double* dbg_ji, double* dbg_jj)
{
    hipStream_t s = h->stream;
    launch_pg_linearize(s, h->n_edge, h->est[h->cur].p, h->ef.p, h->et.p, h->meas.p, h->info, h->opt.huber_delta, h->eb.p, h->econ.p,
                        h->chi_part_lin.p, dbg_e, dbg_ji, dbg_jj);
    launch_pg_gather(s, h->n_blocks, h->g_ptr.p, h->g_ent.p, h->blk_row.p, h->blk_col.p, h->econ.p, h->H.p, h->b.p);
    launch_pg_reduce(s, h->chi_part_lin.p, h->n_chi_part, h->H.p, h->diag_block.p, h->n_free, h->scal.p);
    h->applied_shift = 0.0;   // H is rebuilt
    if (h->two_level && h->cinv_pending) {
        // the coarse inverse under way on the second stream read Ac0 and G: they are rewritten below
        SOSLAM_HIP_CHECK(hipStreamWaitEvent(s, h->ev_coarse_done, 0));
        h->cinv_pending = false;
        h->cinv_cur ^= 1;
    }
    if (h->two_level) {
        launch_pg_coarse_basis(s, h->n_free, h->free_vertex.p, h->row_agg.p, h->agg_ref.p, h->est[h->cur].p, h->cP.p);
        launch_coarse_assemble(s, h->n_agg, h->agg_ptr.p, h->n_cb, h->cb_ptr.p, h->cb_ent.p, h->cb_I.p, h->cb_J.p, h->blk_row.p, h->blk_col.p,
                               h->H.p, h->cP.p, h->ncp, h->cG.p, h->cAc0.p);
    }
    SOSLAM_HIP_CHECK(hipGetLastError());
    return SOSLAM_OK;
}

int read_scal(soslam_pg* h, double* out, int n)
{
    SOSLAM_HIP_CHECK(hipMemcpyAsync(out, h->scal.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    return SOSLAM_OK;
}

float elapsed_ms(soslam_pg* h)
{
    float ms = 0.0f;
    (void)hipEventSynchronize(h->ev[1]);
    (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
    return ms;
}

int run(soslam_pg* h, soslam_pg_summary* out)
{
    if (!h->have_graph) { set_last_error("optimize before set_graph"); return SOSLAM_ERR_STATE; }
    const soslam_pg_options& o = h->opt;
    hipStream_t s = h->stream;
    soslam_pg_summary sum{};
    sum.setup_seconds = h->setup_seconds;
    sum.termination = SOSLAM_PG_TERM_ITERATIONS;
    h->log.clear();
    const double t0 = now_sec();
    double lambda = 0.0, ni = 2.0, current = 0.0;
    double sc[4];
    int it = 0;
    for (it = 0; it < o.max_iterations; it++) {
        SOSLAM_HIP_CHECK(hipEventRecord(h->ev[0], s));
        SOSLAM_CHECK(linearize(h, nullptr, nullptr, nullptr));
        SOSLAM_HIP_CHECK(hipEventRecord(h->ev[1], s));
        SOSLAM_CHECK(read_scal(h, sc, 2));
        sum.linearize_ms += elapsed_ms(h);
        current = sc[0];
        if (it == 0) {
            sum.initial_chi2 = current;
            if (!std::isfinite(current)) { set_last_error("non-finite chi2 at the initial estimates"); return SOSLAM_ERR_NON_FINITE; }
            lambda = o.tau * sc[1];
            ni = 2.0;
        }
        double rho = 0.0;
        int qmax = 0, accepted = 0, lin_total = 0;
        do {
            double rel = 0.0;
            int lin_it = 0;
            SOSLAM_HIP_CHECK(hipEventRecord(h->ev[0], s));
            if (h->n_free && h->band_mode) {
                // the shift into H (the difference to what a previous trial of this iteration left there), the band's factor - which
                // carries b down its tree -, PCG rounds until the tolerance: one round when every block lies inside the band
                if (lambda != h->applied_shift) launch_pg_shift_diag(s, h->H.p, h->diag_block.p, h->n_free, lambda - h->applied_shift);
                h->applied_shift = lambda;
                SOSLAM_HIP_CHECK(hipMemsetAsync(h->lin_scal.p + SC_LIN_ITERS, 0, 3 * sizeof(double), s));
                launch_cr_factor(s, bsr_view(h), h->cr_map.p, h->band_bw, h->cr_ws.p, h->lin_scal.p, nullptr, h->b.p);
                int enq = std::min(o.pcg_max_iterations, h->band_off ? h->band_rounds : 2);
                launch_pcg_cr(s, bsr_view(h), h->band_bw, h->cr_ws.p, h->b.p, h->x.p, h->resid.p, h->work.p, o.pcg_tolerance, enq, h->lin_scal.p, true);
                double lin[3] = {0.0, 0.0, 0.0};
                while (true) {
                    SOSLAM_HIP_CHECK(hipMemcpyAsync(lin, h->lin_scal.p + SC_LIN_ITERS, sizeof lin, hipMemcpyDeviceToHost, s));
                    SOSLAM_HIP_CHECK(hipStreamSynchronize(s));
                    if (lin[1] <= o.pcg_tolerance || lin[2] != 0.0 || enq >= o.pcg_max_iterations) break;
                    const int more = std::min(o.pcg_max_iterations - enq, std::max(4, enq / 2));
                    launch_pcg_cr_more(s, bsr_view(h), h->band_bw, h->cr_ws.p, h->x.p, h->resid.p, h->work.p, o.pcg_tolerance, more, h->lin_scal.p);
                    enq += more;
                }
                static_assert(SC_LIN_RESID == SC_LIN_ITERS + 1 && SC_LIN_STATUS == SC_LIN_ITERS + 2, "the three slots read above");
                lin_it = (lin[2] != 0.0 || !(lin[1] <= o.pcg_tolerance)) ? -1 : (int)lin[0];
                rel = lin[1];
                if (lin_it > 0) h->band_rounds = std::max(2, lin_it + 1);
            } else if (h->n_free && h->two_level) {
                if (h->cinv_pending) {   // a second trial of the same iteration: the inverse the first one started
                    SOSLAM_HIP_CHECK(hipStreamWaitEvent(s, h->ev_coarse_done, 0));
                    h->cinv_pending = false;
                    h->cinv_cur ^= 1;
                }
                if (!h->cinv_valid) {
                    // the first solve of a graph: nothing to lag behind
                    pcg2_coarse_inverse(s, h->n_agg, h->ncp, h->cAc0.p, h->cG.p, lambda, h->cAinv[h->cinv_cur].p, h->cebuf[h->cinv_cur].p, h->cstatus.p);
                    h->cinv_valid = true;
                } else {
                    // this trial's coarse inverse, for the NEXT solve, beside this solve (0.4 - 0.7 ms of small launches that would
                    // otherwise sit in front of every PCG)
                    const int other = h->cinv_cur ^ 1;
                    SOSLAM_HIP_CHECK(hipEventRecord(h->ev_coarse_in, s));
                    SOSLAM_HIP_CHECK(hipStreamWaitEvent(h->cstream, h->ev_coarse_in, 0));
                    pcg2_coarse_inverse(h->cstream, h->n_agg, h->ncp, h->cAc0.p, h->cG.p, lambda, h->cAinv[other].p, h->cebuf[other].p, h->cstatus.p);
                    SOSLAM_HIP_CHECK(hipEventRecord(h->ev_coarse_done, h->cstream));
                    h->cinv_pending = true;
                }
                const TwoLevelView tl{h->n_agg, h->ncp, h->agg_ptr.p, h->cP.p, h->cAinv[h->cinv_cur].p, h->crc.p};
                lin_it = pcg2_solve(s, bsr_view(h), lambda, h->b.p, h->x.p, h->resid.p, h->work.p, tl, o.pcg_tolerance,
                                    o.pcg_max_iterations, h->last_lin_it > 0 ? h->last_lin_it + 2 : 48, &rel);
                if (lin_it > 0) h->last_lin_it = lin_it;
            } else if (h->n_free) {
                lin_it = pcg_multi_solve(s, bsr_view(h), lambda, h->b.p, h->x.p, h->resid.p, h->work.p, o.pcg_tolerance, o.pcg_max_iterations, 32, &rel);
            }
            SOSLAM_HIP_CHECK(hipEventRecord(h->ev[1], s));
            const bool ok = lin_it >= 0;
            lin_total += std::max(lin_it, 0);
            launch_pg_update(s, h->n_vertex, h->est[h->cur].p, h->free_idx.p, h->x.p, h->b.p, lambda, h->est[h->cur ^ 1].p, h->scale_part.p);
            launch_pg_reduce(s, h->scale_part.p, h->n_scale_part, nullptr, nullptr, 0, h->scal.p + 2);
            launch_pg_chi2(s, h->n_edge, h->est[h->cur ^ 1].p, h->ef.p, h->et.p, h->meas.p, h->info, o.huber_delta, h->chi_part.p);
            launch_pg_reduce(s, h->chi_part.p, div_up(h->n_edge, 256), nullptr, nullptr, 0, h->scal.p + 3);
            SOSLAM_HIP_CHECK(hipGetLastError());
            SOSLAM_CHECK(read_scal(h, sc, 4));
            sum.linear_solve_ms += elapsed_ms(h);
            double temp = ok ? sc[3] : 1.7976931348623157e308;
            const double scale = sc[2] + 1e-3;
            rho = (current - temp) / scale;
            if (rho > 0 && std::isfinite(temp)) {
                double alpha = 1.0 - std::pow(2.0 * rho - 1.0, 3.0);
                alpha = std::min(alpha, 2.0 / 3.0);
                lambda *= std::max(1.0 / 3.0, alpha);
                ni = 2.0;
                current = temp;
                h->cur ^= 1;   // discardTop: keep the candidate
                accepted = 1;
            } else {
                lambda *= ni;   // pop: the current estimates stay
                ni *= 2.0;
                if (!std::isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < o.max_trials);
        sum.linear_iterations += lin_total;
        soslam_pg_iteration e{current, lambda, qmax, accepted, lin_total, 0};
        h->log.push_back(e);
        if (o.verbose) std::printf("iteration= %d\t chi2= %.9e\t lambda= %.6e\t levenbergIter= %d\t pcg= %d\n", it, current, lambda, qmax, lin_total);
        if (qmax == o.max_trials || rho == 0 || !std::isfinite(lambda)) {
            sum.termination = SOSLAM_PG_TERM_TRIALS;
            it++;
            break;
        }
    }
    sum.iterations = it;
    sum.final_chi2 = current;
    sum.solve_seconds = now_sec() - t0;
    if (out) *out = sum;
    return SOSLAM_OK;
}

}  // namespace

extern "C" {

void soslam_pg_options_default(soslam_pg_options* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->max_iterations = 10;
    o->max_trials = 10;
    o->huber_delta = 1.0;
    o->tau = 1e-5;
    o->pcg_tolerance = 1e-10;
    o->pcg_max_iterations = 4000;
    o->verbose = 0;
    o->preconditioner = SOSLAM_PG_PRECOND_AUTO;
    o->device = -1;
    o->stream = nullptr;
}

int soslam_pg_create(const soslam_pg_options* opts, soslam_pg** out)
{
    if (!out) return SOSLAM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
        set_last_error("no HIP device visible: the pose-graph backend has no CPU fallback");
        return SOSLAM_ERR_NO_DEVICE;
    }
    std::unique_ptr<soslam_pg> h(new soslam_pg());
    if (opts) h->opt = *opts; else soslam_pg_options_default(&h->opt);
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
    if (h->opt.stream) h->stream = static_cast<hipStream_t>(h->opt.stream);
    else { SOSLAM_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    SOSLAM_HIP_CHECK(hipEventCreate(&h->ev[0]));
    SOSLAM_HIP_CHECK(hipEventCreate(&h->ev[1]));
    *out = h.release();
    return SOSLAM_OK;
}

void soslam_pg_destroy(soslam_pg* h)
{
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    delete h;
}

int soslam_pg_set_graph(soslam_pg* h, uint32_t n_vertex, const double* est, const uint8_t* fixed, uint32_t n_edge,
                        const uint32_t* e_from, const uint32_t* e_to, const double* meas, const double* info36)
{
    if (!h || !n_vertex || !est || !info36 || (n_edge && (!e_from || !e_to || !meas))) return SOSLAM_ERR_INVALID_ARGUMENT;
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    h->have_graph = false;
    return build_graph(h, n_vertex, est, fixed, n_edge, e_from, e_to, meas, info36);
}

int soslam_pg_append(soslam_pg* h, uint32_t n_add_vertex, const double* est_add, const uint8_t* fixed_add, uint32_t n_add_edge,
                     const uint32_t* e_from, const uint32_t* e_to, const double* meas, const double* info36)
{
    if (!h || (n_add_vertex && !est_add) || (n_add_edge && (!e_from || !e_to || !meas))) return SOSLAM_ERR_INVALID_ARGUMENT;
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    if (!h->have_graph) {
        if (!info36 || !n_add_vertex) { set_last_error("the first append needs vertices and the information matrix"); return SOSLAM_ERR_INVALID_ARGUMENT; }
        return build_graph(h, n_add_vertex, est_add, fixed_add, n_add_edge, e_from, e_to, meas, info36);
    }
    // the existing vertices keep the estimates the last optimisation left on the device (as g2o's vertices do)
    std::vector<double> est(h->g_est.size());
    SOSLAM_HIP_CHECK(hipMemcpyAsync(est.data(), h->est[h->cur].p, sizeof(double) * est.size(), hipMemcpyDeviceToHost, h->stream));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    std::vector<double> meas_all(h->g_meas);
    std::vector<uint8_t> fixed(h->g_fixed);
    std::vector<uint32_t> ef(h->g_ef), et(h->g_et);
    est.insert(est.end(), est_add, est_add + 7 * (size_t)n_add_vertex);
    for (uint32_t v = 0; v < n_add_vertex; v++) fixed.push_back(fixed_add ? fixed_add[v] : 0);
    if (n_add_edge) {
        meas_all.insert(meas_all.end(), meas, meas + 7 * (size_t)n_add_edge);
        ef.insert(ef.end(), e_from, e_from + n_add_edge);
        et.insert(et.end(), e_to, e_to + n_add_edge);
    }
    double info[36];
    std::memcpy(info, info36 ? info36 : h->info.m, sizeof info);
    // the block pattern and the gather lists depend on the whole edge set: rebuilt; buffers are kept (grow-only)
    const int st = build_graph(h, (uint32_t)fixed.size(), est.data(), fixed.data(), (uint32_t)ef.size(), ef.data(), et.data(),
                               meas_all.data(), info);
    if (st != SOSLAM_OK) h->have_graph = false;   // a rejected append leaves no half-built graph behind
    return st;
}

int soslam_pg_graph_size(soslam_pg* h, uint32_t* n_vertex, uint32_t* n_edge)
{
    if (!h) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (n_vertex) *n_vertex = h->have_graph ? h->n_vertex : 0;
    if (n_edge) *n_edge = h->have_graph ? h->n_edge : 0;
    return SOSLAM_OK;
}

int soslam_pg_optimize(soslam_pg* h, soslam_pg_summary* summary)
{
    if (!h) return SOSLAM_ERR_INVALID_ARGUMENT;
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    return run(h, summary);
}

int soslam_pg_get_estimates(soslam_pg* h, double* est)
{
    if (!h || !est) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_graph) { set_last_error("get_estimates before set_graph"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    SOSLAM_HIP_CHECK(hipMemcpyAsync(est, h->est[h->cur].p, sizeof(double) * 7 * h->n_vertex, hipMemcpyDeviceToHost, h->stream));
    SOSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    return SOSLAM_OK;
}

int soslam_pg_get_iteration_log(soslam_pg* h, soslam_pg_iteration* out, int32_t capacity, int32_t* count)
{
    if (!h || !count) return SOSLAM_ERR_INVALID_ARGUMENT;
    *count = (int32_t)h->log.size();
    if (out) for (int32_t i = 0; i < capacity && i < *count; i++) out[i] = h->log[(size_t)i];
    return SOSLAM_OK;
}

int soslam_pg_solve(const soslam_pg_options* opts, uint32_t n_vertex, double* est, const uint8_t* fixed, uint32_t n_edge,
                    const uint32_t* e_from, const uint32_t* e_to, const double* meas, const double* info36,
                    soslam_pg_summary* summary)
{
    soslam_pg* h = nullptr;
    SOSLAM_CHECK(soslam_pg_create(opts, &h));
    int st = soslam_pg_set_graph(h, n_vertex, est, fixed, n_edge, e_from, e_to, meas, info36);
    if (st == SOSLAM_OK) st = soslam_pg_optimize(h, summary);
    if (st == SOSLAM_OK) st = soslam_pg_get_estimates(h, est);
    soslam_pg_destroy(h);
    return st;
}

int soslam_pg_debug_linearize(soslam_pg* h, double* edge_e, double* edge_ji, double* edge_jj, double* chi2, double* h_dense,
                              double* b)
{
    if (!h) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_graph) { set_last_error("debug_linearize before set_graph"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    DevBuf<double> de, dji, djj;
    const bool want_edges = edge_e || edge_ji || edge_jj;
    if (want_edges) {
        SOSLAM_CHECK(de.alloc((size_t)h->n_edge * 6));
        SOSLAM_CHECK(dji.alloc((size_t)h->n_edge * 36));
        SOSLAM_CHECK(djj.alloc((size_t)h->n_edge * 36));
    }
    SOSLAM_CHECK(linearize(h, want_edges ? de.p : nullptr, want_edges ? dji.p : nullptr, want_edges ? djj.p : nullptr));
    double sc[2];
    SOSLAM_CHECK(read_scal(h, sc, 2));
    if (chi2) *chi2 = sc[0];
    if (edge_e) SOSLAM_HIP_CHECK(hipMemcpy(edge_e, de.p, sizeof(double) * 6 * h->n_edge, hipMemcpyDeviceToHost));
    if (edge_ji) SOSLAM_HIP_CHECK(hipMemcpy(edge_ji, dji.p, sizeof(double) * 36 * h->n_edge, hipMemcpyDeviceToHost));
    if (edge_jj) SOSLAM_HIP_CHECK(hipMemcpy(edge_jj, djj.p, sizeof(double) * 36 * h->n_edge, hipMemcpyDeviceToHost));
    // the caller's order of the free vertices (vertex order); internally they are numbered aggregate by aggregate (build_graph)
    std::vector<uint32_t> nat(h->n_free);
    {
        uint32_t k = 0;
        for (uint32_t v = 0; v < h->n_vertex; v++)
            if (h->h_free[v] >= 0) nat[(size_t)h->h_free[v]] = k++;
    }
    if (b) {
        std::vector<double> tmp((size_t)h->n_free * 6);
        SOSLAM_HIP_CHECK(hipMemcpy(tmp.data(), h->b.p, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
        for (uint32_t f = 0; f < h->n_free; f++) std::memcpy(b + 6 * (size_t)nat[f], &tmp[6 * (size_t)f], 6 * sizeof(double));
    }
    if (h_dense) {
        const size_t n6 = (size_t)h->n_free * 6;
        std::vector<double> blk((size_t)h->n_blocks * 36);
        SOSLAM_HIP_CHECK(hipMemcpy(blk.data(), h->H.p, sizeof(double) * blk.size(), hipMemcpyDeviceToHost));
        std::memset(h_dense, 0, sizeof(double) * n6 * n6);
        for (uint32_t k = 0; k < h->n_blocks; k++)
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 6; c++) {
                    const size_t i = 6 * (size_t)nat[h->h_blk_row[k]] + a, j = 6 * (size_t)nat[h->h_blk_col[k]] + c;
                    h_dense[i * n6 + j] = blk[36 * (size_t)k + a * 6 + c];
                    h_dense[j * n6 + i] = blk[36 * (size_t)k + a * 6 + c];
                }
    }
    (void)s;
    return SOSLAM_OK;
}

int soslam_pg_time_linearize(soslam_pg* h, int32_t reps, float* avg_ms)
{
    if (!h || reps < 1 || !avg_ms) return SOSLAM_ERR_INVALID_ARGUMENT;
    if (!h->have_graph) { set_last_error("time_linearize before set_graph"); return SOSLAM_ERR_STATE; }
    SOSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    auto once = [&]() {
        launch_pg_linearize(s, h->n_edge, h->est[h->cur].p, h->ef.p, h->et.p, h->meas.p, h->info, h->opt.huber_delta, h->eb.p, h->econ.p,
                            h->chi_part_lin.p, nullptr, nullptr, nullptr);
        launch_pg_gather(s, h->n_blocks, h->g_ptr.p, h->g_ent.p, h->blk_row.p, h->blk_col.p, h->econ.p, h->H.p, h->b.p);
    };
    once();
    SOSLAM_HIP_CHECK(hipEventRecord(h->ev[0], s));
    for (int i = 0; i < reps; i++) once();
    SOSLAM_HIP_CHECK(hipEventRecord(h->ev[1], s));
    *avg_ms = elapsed_ms(h) / (float)reps;
    SOSLAM_HIP_CHECK(hipGetLastError());
    return SOSLAM_OK;
}

}  // extern "C"
