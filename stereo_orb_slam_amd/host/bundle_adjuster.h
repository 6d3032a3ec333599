// bundle_adjuster.h - host shim with the reference's entry point
//   BundleAdjuster(std::vector<Frame*>&, std::vector<MapPoint*>&); void Optimize(unsigned start, unsigned end);
// (/root/reference/src/bundle_adjuster.h:13-25).  Everything ceres::Solve did is behind the C ABI.  The device handle
// lives as long as the adjuster: the per-frame and sliding-window calls of /root/reference/src/slam.cpp:121-129 reuse
// its stream and device allocations instead of creating them once per call.
#pragma once

#include <vector>

#include "camera_frame.h"
#include "map_point.h"
#include "observation.h"
#include "soslam_ba.h"

class BundleAdjuster {
public:
    BundleAdjuster(std::vector<Frame*>& cam_frames, std::vector<MapPoint*>& ldm_points);
    ~BundleAdjuster();
    BundleAdjuster(const BundleAdjuster&) = delete;
    BundleAdjuster& operator=(const BundleAdjuster&) = delete;

    // window [start_frame_id, end_frame_id): first pose of the window constant, every observed point free
    void Optimize(unsigned int start_frame_id, unsigned int end_frame_id);

    // Multi-GPU job, one process per GPU: every rank holds the whole map and calls Optimize() with
    // the same arguments; the window's points are sharded over the ranks (contiguous ranges of the window's first-seen
    // order), the library sums the reduced camera system over the ranks with its own RCCL leg (soslam_ba_init_rccl), and
    // every rank ends up with all optimised poses and points.  rccl_unique_id: the 128 bytes rank 0 drew with
    // soslam_rccl_get_unique_id, handed to the other ranks by the host's side channel.  Call once, before the first
    // Optimize().
    void EnableSharding(int rank, int world, const void* rccl_unique_id);

    // extensions (the reference returns void and prints Ceres' report)
    soslam_ba_options& Options() { return m_options; }
    const soslam_ba_summary& LastSummary() const { return m_summary; }
    int LastStatus() const { return m_status; }

private:
    std::vector<Frame*>& m_cam_frames;
    std::vector<MapPoint*>& m_ldm_points;
    soslam_ba_options m_options;
    soslam_ba_summary m_summary{};
    int m_status = 0;
    soslam_ba* m_handle = nullptr;   // created by the first Optimize()
    int m_rank = 0, m_world = 0;     // m_world > 0: sharded job
    unsigned char m_rccl_id[128] = {0};
    bool m_comm_ready = false;
};
