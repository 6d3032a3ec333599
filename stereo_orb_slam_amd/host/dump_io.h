// dump_io.h - the reference's only on-disk form of a BA problem: VisualOdometer::Dump
// (/root/reference/src/visual_odometer.cpp:446-505) writes three text files into a folder,
//   poses.txt        count, then one line of 16 floats per frame: the camera->WORLD 4x4, row-major
//   points.txt       count, then "x y z" per map point
//   constraints.txt  count, then "frame_id point_id u_l v_l u_r v_r sigma" per observation, frame by frame
// Reading one back rebuilds the Frame / MapPoint containers BundleAdjuster works on, so a dump made by the
// original binary on KITTI feeds this backend unchanged.
#pragma once

#include <cstdio>
#include <fstream>
#include <iomanip>
#include <memory>
#include <string>
#include <vector>

#include "camera_frame.h"
#include "map_point.h"

namespace soslam_host {

struct MapState {
    std::vector<std::unique_ptr<Frame>> frame_store;
    std::vector<std::unique_ptr<MapPoint>> point_store;
    std::vector<Frame*> frames;      // what BundleAdjuster / PoseGraphOptimizer take by reference
    std::vector<MapPoint*> points;
};

inline std::string JoinPath(const std::string& folder, const char* name)
{
    if (folder.empty()) return name;
    return folder.back() == '/' ? folder + name : folder + "/" + name;
}

// Returns false (and leaves `out` empty) if a file is missing or malformed.
inline bool ReadDump(const std::string& folder, MapState& out)
{
    out = MapState{};
    std::ifstream fp(JoinPath(folder, "poses.txt")), fx(JoinPath(folder, "points.txt")), fc(JoinPath(folder, "constraints.txt"));
    if (!fp.is_open() || !fx.is_open() || !fc.is_open()) return false;
    size_t n_pose = 0, n_pt = 0, n_obs = 0;
    if (!(fp >> n_pose) || !(fx >> n_pt) || !(fc >> n_obs)) return false;

    MapState st;
    st.point_store.reserve(n_pt);
    for (size_t i = 0; i < n_pt; i++) {
        float x, y, z;
        if (!(fx >> x >> y >> z)) return false;
        st.point_store.emplace_back(new MapPoint(x, y, z));
        st.points.push_back(st.point_store.back().get());
    }
    st.frame_store.reserve(n_pose);
    for (size_t i = 0; i < n_pose; i++) {
        Mat4f glb;
        for (int e = 0; e < 16; e++)
            if (!(fp >> glb.m[e])) return false;
        // a Frame is built from its pose relative to the previous frame (camera_frame.h); chaining it back
        // reproduces the dumped global pose up to float32 re-orthonormalisation
        Frame* prev = st.frames.empty() ? nullptr : st.frames.back();
        const Mat4f rel = prev ? prev->GlobalPose().inverse() * glb : glb;
        st.frame_store.emplace_back(new Frame(prev, rel));
        st.frames.push_back(st.frame_store.back().get());
    }
    std::vector<char> seen(n_pt, 0);
    for (size_t i = 0; i < n_obs; i++) {
        size_t frame_id, point_id;
        float ul, vl, ur, vr, sigma;
        if (!(fc >> frame_id >> point_id >> ul >> vl >> ur >> vr >> sigma)) return false;
        if (frame_id >= n_pose || point_id >= n_pt) return false;
        Frame* f = st.frames[frame_id];
        f->AddObservation(Observation((unsigned int)point_id, ul, vl, ur, vr, sigma));
        // the dump does not record which frame first observed a point; frames are written in order, so the
        // first frame that lists a point is the one that created it (visual_odometer.cpp:335-444)
        f->AddMapPoints(st.points[point_id], !seen[point_id]);
        seen[point_id] = 1;
    }
    out = std::move(st);
    return true;
}

inline bool WriteDump(const std::string& folder, const std::vector<Frame*>& frames, const std::vector<MapPoint*>& points)
{
    std::ofstream fp(JoinPath(folder, "poses.txt")), fx(JoinPath(folder, "points.txt")), fc(JoinPath(folder, "constraints.txt"));
    if (!fp.is_open() || !fx.is_open() || !fc.is_open()) return false;
    // the reference streams floats at the default precision (6 significant digits); 9 round-trips float32
    fp << std::setprecision(9);
    fx << std::setprecision(9);
    fc << std::setprecision(9);
    fp << frames.size() << "\n";
    for (const Frame* f : frames) {
        const Mat4f t = f->GlobalPose();
        for (int e = 0; e < 16; e++) fp << t.m[e] << (e == 15 ? "\n" : " ");
    }
    fx << points.size() << "\n";
    for (const MapPoint* p : points) {
        const std::array<float, 3> x = p->Position();
        fx << x[0] << " " << x[1] << " " << x[2] << "\n";
    }
    size_t n_obs = 0;
    for (const Frame* f : frames) n_obs += f->Observations().size();
    fc << n_obs << "\n";
    for (size_t i = 0; i < frames.size(); i++)
        for (const Observation& o : frames[i]->Observations())
            fc << i << " " << o.point_id << " " << o.u_l << " " << o.v_l << " " << o.u_r << " " << o.v_r << " " << o.sigma << "\n";
    return fp.good() && fx.good() && fc.good();
}

}  // namespace soslam_host
