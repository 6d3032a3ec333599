// selftest.cpp - sanitizer driver of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT): one small BA solve, one LM step with
// its dense outputs, one sharded step and one pose-graph solve, built with -fsanitize=address,undefined by `make asan`
// (sanitizers on the CPU build only; the GPU pool refuses them).
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/soslam_synth.h"
#include "../stereo_orb_slam_amd/host/mat4f.h"
#include "ba_oracle.h"
#include "pg_oracle.h"

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) { std::fprintf(stderr, "FAILED: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } \
    } while (0)

int main()
{
    // ---- bundle adjustment: 8 cameras, 300 points, geometric tracks
    soslam_synth_ba_params p;
    CHECK(soslam_synth_ba_config(1, &p) == 0);
    p.n_cam = 8; p.n_pt = 300; p.track_len = 4;
    uint32_t n_obs = 0;
    CHECK(soslam_synth_ba_count(&p, &n_obs) == 0 && n_obs > 0);
    std::vector<float> poses((size_t)p.n_cam * 16), pts((size_t)p.n_pt * 3), uv((size_t)n_obs * 4);
    std::vector<uint32_t> oc(n_obs), op(n_obs);
    double pl[12], pr[12];
    CHECK(soslam_synth_ba_generate(&p, poses.data(), pts.data(), oc.data(), op.data(), uv.data(), pl, pr, nullptr, nullptr) == 0);
    std::vector<double> cams((size_t)p.n_cam * 6), x((size_t)p.n_pt * 3);
    for (uint32_t c = 0; c < p.n_cam; c++) {
        soslam_host::Mat4f m;
        for (int i = 0; i < 16; i++) m.m[i] = poses[16 * (size_t)c + i];
        std::array<double, 6> q;
        soslam_host::MatrixToPose(m.inverse(), q);
        for (int i = 0; i < 6; i++) cams[6 * (size_t)c + i] = q[i];
    }
    for (size_t i = 0; i < x.size(); i++) x[i] = pts[i];
    std::vector<uint8_t> fixed(p.n_cam, 0);
    fixed[0] = 1;
    oracle_ba_options o;
    oracle_ba_options_default(&o);
    o.max_iterations = 15; o.num_threads = 2;
    const size_t n6 = 6 * (size_t)(p.n_cam - 1);
    std::vector<double> S(n6 * n6), rhs(n6), dc((size_t)p.n_cam * 6), dp(x.size()), sc(4);
    CHECK(oracle_ba_step(p.n_cam, p.n_pt, n_obs, oc.data(), op.data(), uv.data(), cams.data(), x.data(), pl, pr, fixed.data(), &o, 1e4,
                         S.data(), rhs.data(), dc.data(), dp.data(), sc.data()) == 0);
    CHECK(sc[1] > 0.0 && std::isfinite(sc[2]));
    CHECK(oracle_ba_step_sharded(3, p.n_cam, p.n_pt, n_obs, oc.data(), op.data(), uv.data(), cams.data(), x.data(), pl, pr, fixed.data(), &o,
                                 1e4, S.data(), rhs.data()) == 0);
    oracle_ba_summary sum;
    std::vector<oracle_ba_iteration> log((size_t)o.max_iterations + 1);
    CHECK(oracle_ba_solve(p.n_cam, p.n_pt, n_obs, oc.data(), op.data(), uv.data(), cams.data(), x.data(), pl, pr, fixed.data(), &o, &sum,
                          log.data()) == 0);
    CHECK(sum.final_cost < sum.initial_cost && sum.iterations >= 1);
    std::printf("ba: %u obs, cost %.6e -> %.6e in %d iterations\n", n_obs, sum.initial_cost, sum.final_cost, sum.iterations);

    // ---- pose graph: 200 nodes
    soslam_synth_pg_params g;
    CHECK(soslam_synth_pg_config(6, &g) == 0);
    uint32_t n_edge = 0;
    CHECK(soslam_synth_pg_count(&g, &n_edge) == 0 && n_edge > 0);
    std::vector<double> est((size_t)g.n_node * 7), meas((size_t)n_edge * 7);
    std::vector<uint32_t> ef(n_edge), et(n_edge);
    CHECK(soslam_synth_pg_generate(&g, est.data(), ef.data(), et.data(), meas.data(), nullptr) == 0);
    std::vector<uint8_t> vfixed(g.n_node, 0);
    vfixed[0] = 1;
    double info[36] = {0};
    for (int i = 0; i < 6; i++) info[i * 7] = i < 3 ? 0.01 : 1.0;
    oracle_pg_options po;
    oracle_pg_options_default(&po);
    po.max_iterations = 4; po.num_threads = 2;
    oracle_pg_summary ps;
    std::vector<oracle_pg_iteration> plog((size_t)po.max_iterations + 1);
    CHECK(oracle_pg_solve(g.n_node, n_edge, est.data(), vfixed.data(), ef.data(), et.data(), meas.data(), info, &po, &ps, plog.data()) == 0);
    CHECK(ps.final_chi2 < ps.initial_chi2);
    std::printf("pg: %u edges, chi2 %.6e -> %.6e\nOK\n", n_edge, ps.initial_chi2, ps.final_chi2);
    return 0;
}
