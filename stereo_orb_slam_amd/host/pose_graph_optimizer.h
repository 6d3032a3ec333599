// pose_graph_optimizer.h - host shim with the reference's entry point
//   PoseGraphOptimizer(BundleAdjuster&, std::vector<Frame*>&, std::vector<PoseGraphEdge>&); void Optimize();
// (/root/reference/src/pose_graph_optimizer.h:18-50).  The g2o graph the reference keeps inside the object
// (vertices, edges, m_last_id; it persists and grows between calls, pose_graph_optimizer.cpp:56-59) lives in ONE device
// handle for the object's lifetime: every Optimize() appends the new vertices and edges (soslam_pg_append), the
// estimates of the existing vertices stay on the device between calls as g2o's do.  The arrays below mirror it for
// SavePoseGraph and the write-back.
#pragma once

#include <array>
#include <map>
#include <string>
#include <vector>

#include "bundle_adjuster.h"
#include "camera_frame.h"
#include "pose_graph.h"
#include "soslam_pg.h"

class PoseGraphOptimizer {
public:
    PoseGraphOptimizer(BundleAdjuster& ba, std::vector<Frame*>& cam_frames, std::vector<PoseGraphEdge>& edges);
    ~PoseGraphOptimizer();
    PoseGraphOptimizer(const PoseGraphOptimizer&) = delete;
    PoseGraphOptimizer& operator=(const PoseGraphOptimizer&) = delete;

    void Optimize();

    // The reference derives a loop edge's relative pose with its image front-end (descriptor matching + RANSAC,
    // /root/reference/src/pose_graph_optimizer.cpp:175-249), which is out of scope here: the measurement arrives as
    // data.  trans = the 4x4 the reference's CalcTransformation returns for the pair (id_1, id_2); the edge goes
    // from vertex id_2 to vertex id_1 (:193-198).  A loop edge without a registered measurement is skipped, like
    // a failed RANSAC in the reference (:190).
    void AddLoopMeasurement(int id_1, int id_2, const soslam_host::Mat4f& trans);

    // extensions
    soslam_pg_options& Options() { return m_options; }
    const soslam_pg_summary& LastSummary() const { return m_summary; }
    int LastStatus() const { return m_status; }
    void RunGlobalBA(bool on) { m_run_ba = on; }                 // the trailing m_ba.Optimize(0, size-1) (:95)
    bool SavePoseGraph(const std::string& file_path) const;      // text format of :251-286

private:
    void AddOdometryConstraints(unsigned int start_frame_id, unsigned int end_frame_id);
    void AddLoopClosureConstraints();

    BundleAdjuster& m_ba;
    std::vector<Frame*>& m_cam_frames;
    std::vector<PoseGraphEdge>& m_loop_edges;

    std::vector<std::array<double, 7>> m_vertices;   // estimate of frame i: tx ty tz qx qy qz qw
    std::vector<unsigned char> m_fixed;
    std::vector<uint32_t> m_edge_from, m_edge_to;
    std::vector<std::array<double, 7>> m_edge_meas;
    std::map<std::pair<int, int>, soslam_host::Mat4f> m_loop_meas;
    std::array<double, 36> m_information{};
    unsigned int m_last_id = 0;
    bool m_has_first = false;
    bool m_run_ba = true;
    soslam_pg_options m_options;
    soslam_pg_summary m_summary{};
    int m_status = 0;
    soslam_pg* m_pg = nullptr;        // the device-resident graph, created at the first Optimize()
    size_t m_sent_vertices = 0;       // how much of m_vertices / m_edge_* the handle already holds
    size_t m_sent_edges = 0;
};
