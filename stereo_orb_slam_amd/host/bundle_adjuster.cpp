// bundle_adjuster.cpp - gather / solve / write-back around the C ABI.
//
// Conventions kept from /root/reference/src/bundle_adjuster.cpp:39-133:
//   - half-open frame range [start, end) (:62);
//   - a pose enters as the WORLD->CAMERA transform: GlobalPose().inverse() -> MatrixToPose (:66-69);
//   - points are deduplicated in first-seen order over the window (:79-95), float32 -> double;
//   - the first pose of the window is constant (:113); bounds +-1e4 on every point coordinate (:104-108);
//   - results: poses first (PoseToMatrix -> inverse -> Frame::GlobalPose, which moves first-observed
//     points), then every window point is overwritten with its optimised position (:120-132).
#include "bundle_adjuster.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include "mat4f.h"
#include "params.h"
#include "reprojection_error.h"

BundleAdjuster::BundleAdjuster(std::vector<Frame*>& cam_frames, std::vector<MapPoint*>& ldm_points)
    : m_cam_frames(cam_frames), m_ldm_points(ldm_points)
{
    soslam_ba_options_default(&m_options);
    m_options.max_iterations = (int32_t)BA_MAX_ITERATION;
    m_options.lower_bound = BA_POINT_COORD_LOWER_BOUND;
    m_options.upper_bound = BA_POINT_COORD_UPPER_BOUND;
    m_options.huber_delta = 1.0;
    m_options.function_tolerance = 1e-16;
    m_options.gradient_tolerance = 1e-16;
    m_options.verbose = 1;                      // minimizer_progress_to_stdout = true (:15)
    m_options.max_solver_time_seconds = 0.0;    // set to BA_MAX_TIME_SEC for the stock wall-clock cap (:18)
}

BundleAdjuster::~BundleAdjuster() { soslam_ba_destroy(m_handle); }

void BundleAdjuster::EnableSharding(int rank, int world, const void* rccl_unique_id)
{
    m_rank = rank; m_world = world;
    std::memcpy(m_rccl_id, rccl_unique_id, sizeof m_rccl_id);
    m_comm_ready = false;
}

void BundleAdjuster::Optimize(unsigned int start_frame_id, unsigned int end_frame_id)
{
    m_status = SOSLAM_OK;
    m_summary = soslam_ba_summary{};
    if (end_frame_id > m_cam_frames.size()) end_frame_id = (unsigned int)m_cam_frames.size();
    if (start_frame_id >= end_frame_id) return;
    const uint32_t n_cam = end_frame_id - start_frame_id;

    std::vector<double> poses((size_t)n_cam * 6);
    std::vector<double> points;
    std::vector<unsigned int> point_ids;
    std::unordered_map<unsigned int, uint32_t> glb_to_local;
    std::vector<uint32_t> obs_cam, obs_pt;
    std::vector<float> obs_uv;

    for (uint32_t c = 0; c < n_cam; c++) {
        const Frame* frame = m_cam_frames[start_frame_id + c];
        std::array<double, 6> pose;
        soslam_host::MatrixToPose(frame->GlobalPose().inverse(), pose);
        for (int a = 0; a < 6; a++) poses[6 * (size_t)c + a] = pose[a];
        for (const Observation& obs : frame->Observations()) {
            auto it = glb_to_local.find(obs.point_id);
            uint32_t local;
            if (it == glb_to_local.end()) {
                local = (uint32_t)point_ids.size();
                glb_to_local.emplace(obs.point_id, local);
                point_ids.push_back(obs.point_id);
                const std::array<float, 3> p = m_ldm_points[obs.point_id]->Position();
                points.push_back(p[0]); points.push_back(p[1]); points.push_back(p[2]);
            } else {
                local = it->second;
            }
            obs_cam.push_back(c);
            obs_pt.push_back(local);
            obs_uv.push_back(obs.u_l); obs_uv.push_back(obs.v_l); obs_uv.push_back(obs.u_r); obs_uv.push_back(obs.v_r);
        }
    }
    std::vector<uint8_t> fixed(n_cam, 0);
    fixed[0] = 1;

    // create / upload / solve / download on the adjuster's own handle; poses and points are local copies, so the map
    // is only touched after every step has succeeded
    m_status = m_handle ? soslam_ba_set_options(m_handle, &m_options) : soslam_ba_create(&m_options, &m_handle);
    if (m_status == SOSLAM_OK) m_status = soslam_ba_set_projection(m_handle, ReprojectionError::p_l.data(), ReprojectionError::p_r.data());
    if (m_world > 0) {
        // ---- sharded job: this rank's contiguous share of the window's points, every camera, the job-wide block pattern
        if (m_status == SOSLAM_OK && !m_comm_ready) {
            m_status = soslam_ba_init_rccl(m_handle, m_rccl_id, m_rank, m_world);   // collective over the ranks
            m_comm_ready = m_status == SOSLAM_OK;
        }
        const uint32_t n_pt = (uint32_t)point_ids.size();
        uint32_t pb = 0, pe = 0;
        soslam_ba_shard_range(n_pt, m_rank, m_world, &pb, &pe);
        std::vector<uint32_t> s_cam, s_pt;
        std::vector<float> s_uv;
        for (size_t k = 0; k < obs_cam.size(); k++) {
            if (obs_pt[k] < pb || obs_pt[k] >= pe) continue;
            s_cam.push_back(obs_cam[k]);
            s_pt.push_back(obs_pt[k] - pb);
            s_uv.insert(s_uv.end(), obs_uv.begin() + 4 * (std::ptrdiff_t)k, obs_uv.begin() + 4 * (std::ptrdiff_t)k + 4);
        }
        // camera pairs that share a point anywhere in the job: every rank lays the reduced system out alike
        std::vector<std::vector<uint32_t>> cams_of(n_pt);
        for (size_t k = 0; k < obs_cam.size(); k++) cams_of[obs_pt[k]].push_back(obs_cam[k]);
        std::vector<uint64_t> keys;
        for (const auto& cl : cams_of)
            for (size_t a = 0; a < cl.size(); a++)
                for (size_t b = a + 1; b < cl.size(); b++)
                    if (cl[a] != cl[b]) keys.push_back(((uint64_t)std::min(cl[a], cl[b]) << 32) | std::max(cl[a], cl[b]));
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        std::vector<uint32_t> pa(keys.size()), pbv(keys.size());
        for (size_t i = 0; i < keys.size(); i++) { pa[i] = (uint32_t)(keys[i] >> 32); pbv[i] = (uint32_t)(keys[i] & 0xFFFFFFFFu); }
        if (m_status == SOSLAM_OK) m_status = soslam_ba_set_covisibility(m_handle, keys.size(), pa.data(), pbv.data());
        if (m_status == SOSLAM_OK)
            m_status = soslam_ba_set_problem(m_handle, n_cam, pe - pb, (uint32_t)s_cam.size(), s_cam.data(), s_pt.data(), s_uv.data(),
                                             fixed.data());
        if (m_status == SOSLAM_OK) m_status = soslam_ba_set_state(m_handle, poses.data(), points.data() + 3 * (size_t)pb);
        // A rank whose set-up failed (a rank-specific limit, a malformed shard) must not leave the others waiting in the solve's
        // first all-reduce: the ranks agree on one status word and leave together.  Needs the communicator (m_comm_ready).
        auto agree = [&]() {
            if (!m_comm_ready) return;
            int all = m_status;
            const int st = soslam_ba_agree_status(m_handle, m_status, &all);
            if (m_status == SOSLAM_OK && (st != SOSLAM_OK || all != SOSLAM_OK)) m_status = st != SOSLAM_OK ? st : SOSLAM_ERR_COMM;   // a peer failed
        };
        agree();
        if (m_status == SOSLAM_OK) m_status = soslam_ba_solve(m_handle, &m_summary);
        agree();
        // poses are replicated; the points of all ranks come together with one all-reduce
        if (m_status == SOSLAM_OK) m_status = soslam_ba_get_state_global(m_handle, poses.data(), n_pt, pb, points.data());
    } else {
        if (m_status == SOSLAM_OK)
            m_status = soslam_ba_set_problem(m_handle, n_cam, (uint32_t)point_ids.size(), (uint32_t)obs_cam.size(), obs_cam.data(),
                                             obs_pt.data(), obs_uv.data(), fixed.data());
        if (m_status == SOSLAM_OK) m_status = soslam_ba_set_state(m_handle, poses.data(), points.data());
        if (m_status == SOSLAM_OK) m_status = soslam_ba_solve(m_handle, &m_summary);
        if (m_status == SOSLAM_OK) m_status = soslam_ba_get_state(m_handle, poses.data(), points.data());
    }
    if (m_status != SOSLAM_OK) {
        // the caller's map state is left untouched on failure (failure handling)
        std::fprintf(stderr, "[FAIL]: bundle adjustment failed: %s (%s)\n", soslam_status_string(m_status), soslam_last_error());
        return;
    }
    if (m_options.verbose)
        std::printf("[INFO]: BA %u frames / %zu points / %zu observations: cost %.6e -> %.6e in %d iterations (%d accepted), %.3f ms\n",
                    n_cam, point_ids.size(), obs_cam.size(), m_summary.initial_cost, m_summary.final_cost, m_summary.iterations,
                    m_summary.accepted, 1e3 * m_summary.solve_seconds);

    for (uint32_t c = 0; c < n_cam; c++) {
        std::array<double, 6> pose;
        for (int a = 0; a < 6; a++) pose[a] = poses[6 * (size_t)c + a];
        soslam_host::Mat4f t_cw;
        soslam_host::PoseToMatrix(pose, t_cw);
        m_cam_frames[start_frame_id + c]->GlobalPose(t_cw.inverse());
    }
    for (size_t i = 0; i < point_ids.size(); i++)
        m_ldm_points[point_ids[i]]->Position(std::array<double, 3>{points[3 * i], points[3 * i + 1], points[3 * i + 2]});
}
