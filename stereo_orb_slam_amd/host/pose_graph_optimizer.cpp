// pose_graph_optimizer.cpp - graph bookkeeping around the pose-graph C ABI.
//
// Kept from /root/reference/src/pose_graph_optimizer.cpp:
//   - the frame range: all frames but the last when there is no loop edge, else up to the largest loop id (:37-53);
//   - one VertexSE3 per frame from its float32 GlobalPose as [t, q] (:103-116,131-139), vertex 0 fixed (:118-121);
//   - one EdgeSE3 (i-1 -> i) per new frame, measurement = the frame's float32 RelativePose (:149-161);
//   - loop edges from vertex id_2 to vertex id_1 (:193-198); m_loop_edges is cleared afterwards (:216);
//   - information diag(.01,.01,.01,1,1,1), translation rows first (:23-26); Huber kernel; 10 iterations (:69);
//   - write-back [t, q] -> Matrix4f -> Frame::GlobalPose for every vertex (:72-86), UpdatePose() for the frames
//     beyond the graph (:89-92), then the global bundle adjustment over [0, size-1) (:95).
// Consciously not reproduced: the reference re-adds the vertex of frame m_last_id on every later call (:103-124),
// which g2o rejects as a duplicate id while the pointer is still appended to m_vertices; here every frame has
// exactly one vertex.
#include "pose_graph_optimizer.h"

#include <cstdio>
#include <fstream>
#include <iomanip>

#include "params.h"

using soslam_host::Mat4f;

static std::array<double, 7> ToVertex(const Mat4f& pose)
{
    const soslam_host::Quatf q = soslam_host::QuatFromRotation(pose);   // Eigen::Quaternionf(R), not normalised
    return {pose(0, 3), pose(1, 3), pose(2, 3), q.x, q.y, q.z, q.w};
}

PoseGraphOptimizer::PoseGraphOptimizer(BundleAdjuster& ba, std::vector<Frame*>& cam_frames, std::vector<PoseGraphEdge>& edges)
    : m_ba(ba), m_cam_frames(cam_frames), m_loop_edges(edges)
{
    soslam_pg_options_default(&m_options);
    m_options.max_iterations = (int32_t)PG_NUM_ITERATIONS;
    m_options.verbose = 1;   // m_optimizer.setVerbose(true) (:21)
    for (int i = 0; i < 6; i++) m_information[i * 7] = i < 3 ? 0.01 : 1.0;
}

PoseGraphOptimizer::~PoseGraphOptimizer() { soslam_pg_destroy(m_pg); }

void PoseGraphOptimizer::AddLoopMeasurement(int id_1, int id_2, const Mat4f& trans) { m_loop_meas[{id_1, id_2}] = trans; }

void PoseGraphOptimizer::AddOdometryConstraints(unsigned int start_frame_id, unsigned int end_frame_id)
{
    if (!m_has_first) {
        m_vertices.push_back(ToVertex(m_cam_frames[start_frame_id]->GlobalPose()));
        m_fixed.push_back(start_frame_id == 0 ? 1 : 0);
        m_has_first = true;
    }
    for (unsigned int i = start_frame_id + 1; i <= end_frame_id; i++) {
        const Frame* frame = m_cam_frames[i];
        m_vertices.push_back(ToVertex(frame->GlobalPose()));
        m_fixed.push_back(0);
        m_edge_from.push_back(i - 1);
        m_edge_to.push_back(i);
        m_edge_meas.push_back(ToVertex(frame->RelativePose()));
    }
}

void PoseGraphOptimizer::AddLoopClosureConstraints()
{
    for (const PoseGraphEdge& edge : m_loop_edges) {
        const auto it = m_loop_meas.find({edge.first, edge.second});
        if (it == m_loop_meas.end()) continue;
        if (edge.first < 0 || edge.second < 0 || (size_t)edge.first >= m_vertices.size() || (size_t)edge.second >= m_vertices.size()) continue;
        m_edge_from.push_back((uint32_t)edge.second);
        m_edge_to.push_back((uint32_t)edge.first);
        m_edge_meas.push_back(ToVertex(it->second));
    }
    m_loop_edges.clear();
    m_loop_meas.clear();
}

void PoseGraphOptimizer::Optimize()
{
    m_status = SOSLAM_OK;
    m_summary = soslam_pg_summary{};
    if (m_cam_frames.empty()) return;
    unsigned int end_id = 0;
    if (m_loop_edges.empty()) {
        end_id = (unsigned int)m_cam_frames.size() - 1;
    } else {
        for (const PoseGraphEdge& e : m_loop_edges) end_id = std::max(end_id, (unsigned int)std::max(e.second, 0));
        end_id = std::min(end_id, (unsigned int)m_cam_frames.size() - 1);
    }
    if (end_id >= m_last_id) AddOdometryConstraints(m_last_id, end_id);
    AddLoopClosureConstraints();
    m_last_id = std::max(m_last_id, end_id);

    // gauge: with no fixed vertex the reference fixes the one g2o's findGauge() returns (:61-65)
    bool any_fixed = false;
    for (unsigned char f : m_fixed) any_fixed = any_fixed || f;
    if (!any_fixed && !m_fixed.empty()) m_fixed[0] = 1;

    // the device-resident graph: created once, then only what is new since the last call is appended
    if (!m_pg) m_status = soslam_pg_create(&m_options, &m_pg);
    if (m_status == SOSLAM_OK) {
        const size_t nv = m_vertices.size() - m_sent_vertices, ne = m_edge_from.size() - m_sent_edges;
        std::vector<double> est_add(nv * 7), meas_add(ne * 7);
        for (size_t i = 0; i < nv; i++) for (int a = 0; a < 7; a++) est_add[7 * i + a] = m_vertices[m_sent_vertices + i][a];
        for (size_t i = 0; i < ne; i++) for (int a = 0; a < 7; a++) meas_add[7 * i + a] = m_edge_meas[m_sent_edges + i][a];
        if (nv || ne)
            m_status = soslam_pg_append(m_pg, (uint32_t)nv, nv ? est_add.data() : nullptr, nv ? m_fixed.data() + m_sent_vertices : nullptr,
                                        (uint32_t)ne, ne ? m_edge_from.data() + m_sent_edges : nullptr,
                                        ne ? m_edge_to.data() + m_sent_edges : nullptr, ne ? meas_add.data() : nullptr, m_information.data());
        if (m_status == SOSLAM_OK) { m_sent_vertices = m_vertices.size(); m_sent_edges = m_edge_from.size(); }
    }
    std::vector<double> est(m_vertices.size() * 7);
    if (m_status == SOSLAM_OK) m_status = soslam_pg_optimize(m_pg, &m_summary);
    if (m_status == SOSLAM_OK) m_status = soslam_pg_get_estimates(m_pg, est.data());
    if (m_status != SOSLAM_OK) {
        std::fprintf(stderr, "[FAIL]: pose graph optimisation failed: %s (%s)\n", soslam_status_string(m_status), soslam_last_error());
        return;
    }
    for (size_t i = 0; i < m_vertices.size(); i++) for (int a = 0; a < 7; a++) m_vertices[i][a] = est[7 * i + a];

    // update: Quaternionf(w, x, y, z).toRotationMatrix() in float32 (:77-85)
    for (size_t i = 0; i < m_vertices.size() && i < m_cam_frames.size(); i++) {
        const std::array<double, 7>& z = m_vertices[i];
        Mat4f pose = Mat4f::Identity();
        soslam_host::SetRotation(pose, soslam_host::Quatf{(float)z[3], (float)z[4], (float)z[5], (float)z[6]});
        pose(0, 3) = (float)z[0]; pose(1, 3) = (float)z[1]; pose(2, 3) = (float)z[2];
        m_cam_frames[i]->GlobalPose(pose);
    }
    // propagate
    for (size_t i = m_vertices.size(); i < m_cam_frames.size(); i++) m_cam_frames[i]->UpdatePose();
    // global ba
    if (m_run_ba && m_cam_frames.size() > 1) m_ba.Optimize(0, (unsigned int)m_cam_frames.size() - 1);
}

bool PoseGraphOptimizer::SavePoseGraph(const std::string& file_path) const
{
    std::ofstream out(file_path);
    if (!out.is_open()) return false;
    out << std::setprecision(17);
    out << m_vertices.size() << " " << m_edge_from.size() << std::endl;
    for (const auto& v : m_vertices) out << v[0] << " " << v[1] << " " << v[2] << " " << v[3] << " " << v[4] << " " << v[5] << " " << v[6] << "\n";
    for (size_t i = 0; i < m_edge_from.size(); i++) {
        const auto& z = m_edge_meas[i];
        out << m_edge_from[i] << " " << m_edge_to[i] << " " << z[0] << " " << z[1] << " " << z[2] << " " << z[3] << " " << z[4] << " "
            << z[5] << " " << z[6] << "\n";
    }
    return out.good();
}
