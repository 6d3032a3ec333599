// map_point.h - landmark of the host shim: float32 position, no descriptor storage.
// Interface of /root/reference/src/map_point.h:18-46 minus the cv::Mat descriptor members (the image
// front-end is out of scope).
#pragma once

#include <array>

#include "mat4f.h"

class MapPoint {
public:
    MapPoint(float x, float y, float z) : m_position{x, y, z} {}

    // homogeneous transform in float32, like Eigen::Matrix4f * Vector4f
    void Transform(const soslam_host::Mat4f& t)
    {
        const float x = m_position[0], y = m_position[1], z = m_position[2];
        float out[3];
        for (int i = 0; i < 3; i++) out[i] = t(i, 0) * x + t(i, 1) * y + t(i, 2) * z + t(i, 3) * 1.0f;
        m_position = {out[0], out[1], out[2]};
    }

    // the optimiser writes doubles back; they are narrowed to float32 here (/root/reference/src/map_point.h:30-35)
    void Position(std::array<double, 3> pos) { m_position = {(float)pos[0], (float)pos[1], (float)pos[2]}; }
    void Position(const std::array<float, 3>& pos) { m_position = pos; }
    std::array<float, 3> Position() const { return m_position; }

private:
    std::array<float, 3> m_position;
};
