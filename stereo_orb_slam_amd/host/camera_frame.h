// camera_frame.h - keyframe of the host shim: global / relative pose chain, observations, map-point refs.
//
// Behaviour restated from /root/reference/src/camera_frame.h:11-136 with Eigen/OpenCV removed.  The pose
// setters keep their side effects because BundleAdjuster / PoseGraphOptimizer write results through them:
//   GlobalPose(T):   points FIRST OBSERVED in this frame move by T * old^-1; the stored pose is
//                    re-orthonormalised; the relative pose to the previous frame's CURRENT global pose is
//                    recomputed (:32-49)
//   RelativePose(T): global pose re-chained from the previous frame; first-observed points move by
//                    old * new^-1 - the operand order differs from GlobalPose() in the reference (:68) and
//                    is preserved here, not "fixed"
//   UpdatePose():    RelativePose(current relative pose) (:72)
#pragma once

#include <vector>

#include "map_point.h"
#include "mat4f.h"
#include "observation.h"

class Frame {
public:
    using Mat4f = soslam_host::Mat4f;

    Frame(Frame* prev_frame, const Mat4f& pose_rel) : m_prev_frame(prev_frame), m_pose_rel(pose_rel)
    {
        soslam_host::Normalize(m_pose_rel);
        if (m_prev_frame) {
            m_pose_glb = m_prev_frame->GlobalPose() * m_pose_rel;
            soslam_host::Normalize(m_pose_glb);
        } else {
            m_pose_glb = m_pose_rel;
        }
    }

    Mat4f GlobalPose() const { return m_pose_glb; }
    Mat4f RelativePose() const { return m_pose_rel; }

    void GlobalPose(const Mat4f& pose)
    {
        TransformMapPoints(pose * m_pose_glb.inverse());
        m_pose_glb = pose;
        soslam_host::Normalize(m_pose_glb);
        const Mat4f prev = m_prev_frame ? m_prev_frame->GlobalPose() : Mat4f::Identity();
        m_pose_rel = prev.inverse() * m_pose_glb;
        soslam_host::Normalize(m_pose_rel);
    }

    void RelativePose(const Mat4f& pose)
    {
        m_pose_rel = pose;
        soslam_host::Normalize(m_pose_rel);
        const Mat4f prev = m_prev_frame ? m_prev_frame->GlobalPose() : Mat4f::Identity();
        const Mat4f old_glb = m_pose_glb;
        m_pose_glb = prev * m_pose_rel;
        soslam_host::Normalize(m_pose_glb);
        TransformMapPoints(old_glb * m_pose_glb.inverse());
    }

    void UpdatePose() { RelativePose(m_pose_rel); }

    void AddObservation(const Observation& obs) { m_observations.emplace_back(obs); }
    // the reference returns a copy (:75); a const reference is the same data without the allocation
    const std::vector<Observation>& Observations() const { return m_observations; }

    MapPoint* MapPointRef(int idx) { return m_point_refs[idx]; }
    std::vector<MapPoint*> MapPointRefs() { return m_point_refs; }

    void AddMapPoints(MapPoint* point, bool first_observed)
    {
        m_point_refs.emplace_back(point);
        m_point_first.emplace_back(first_observed);
    }

    // loop closure re-targets an observation to an existing map point (:92-109; descriptors not kept here)
    void UpdateMapPoint(int idx, int point_id, MapPoint* point_ref, bool first_observed)
    {
        m_observations[idx].point_id = point_id;
        m_point_refs[idx] = point_ref;
        m_point_first[idx] = first_observed;
    }

    void TransformMapPoints(const Mat4f& trans)
    {
        for (size_t i = 0; i < m_point_refs.size(); i++)
            if (m_point_first[i]) m_point_refs[i]->Transform(trans);
    }

private:
    Frame* m_prev_frame = nullptr;
    Mat4f m_pose_glb;
    Mat4f m_pose_rel;
    std::vector<MapPoint*> m_point_refs;   // map points seen from this frame
    std::vector<char> m_point_first;       // 1 if the point was first observed in this frame
    std::vector<Observation> m_observations;
};
