// mat4f.h - dependency-free float32 4x4 / quaternion / angle-axis helpers for the host shim.
//
// The reference keeps poses as Eigen::Matrix4f and converts through Eigen::AngleAxisf / Quaternionf
// (/root/reference/src/math_utils.h:6-41).  Eigen is not available here, so the same operations are
// restated in float32 arithmetic following Eigen's documented algorithms (rotation matrix -> quaternion by
// Shepperd's branches, quaternion -> angle-axis by atan2 of the vector norm).  Bit-equality with Eigen is
// not claimed and cannot be checked in this image; results agree to float32 rounding.
#pragma once

#include <array>
#include <cmath>
#include <cstring>

namespace soslam_host {

struct Mat4f {
    float m[16];  // row-major

    static Mat4f Identity()
    {
        Mat4f r;
        std::memset(r.m, 0, sizeof r.m);
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
        return r;
    }
    float& operator()(int i, int j) { return m[i * 4 + j]; }
    float operator()(int i, int j) const { return m[i * 4 + j]; }

    Mat4f operator*(const Mat4f& o) const
    {
        Mat4f r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                float s = 0.0f;
                for (int k = 0; k < 4; k++) s += m[i * 4 + k] * o.m[k * 4 + j];
                r.m[i * 4 + j] = s;
            }
        return r;
    }

    // general 4x4 inverse by cofactors (the reference calls Matrix4f::inverse(), not a rigid shortcut)
    Mat4f inverse() const
    {
        const float* a = m;
        float inv[16];
        inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
        inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
        inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
        inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
        inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
        inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
        inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
        inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
        inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
        inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
        inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
        inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
        inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
        inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
        inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
        inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
        const float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
        const float id = 1.0f / det;
        Mat4f r;
        for (int i = 0; i < 16; i++) r.m[i] = inv[i] * id;
        return r;
    }
};

struct Quatf {
    float x, y, z, w;
};

// Quaternionf(R): Shepperd's method, the branch order Eigen uses
inline Quatf QuatFromRotation(const Mat4f& t)
{
    Quatf q;
    float tr = t(0, 0) + t(1, 1) + t(2, 2);
    if (tr > 0.0f) {
        float s = std::sqrt(tr + 1.0f);
        q.w = 0.5f * s;
        s = 0.5f / s;
        q.x = (t(2, 1) - t(1, 2)) * s;
        q.y = (t(0, 2) - t(2, 0)) * s;
        q.z = (t(1, 0) - t(0, 1)) * s;
    } else {
        int i = 0;
        if (t(1, 1) > t(0, 0)) i = 1;
        if (t(2, 2) > t(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        float s = std::sqrt(t(i, i) - t(j, j) - t(k, k) + 1.0f);
        float v[3];
        v[i] = 0.5f * s;
        s = 0.5f / s;
        q.w = (t(k, j) - t(j, k)) * s;
        v[j] = (t(j, i) + t(i, j)) * s;
        v[k] = (t(k, i) + t(i, k)) * s;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

inline Quatf Normalized(Quatf q)
{
    const float n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    if (n > 0.0f) { q.x /= n; q.y /= n; q.z /= n; q.w /= n; }
    return q;
}

inline void SetRotation(Mat4f& t, const Quatf& q)
{
    const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    t(0, 0) = 1.0f - (tyy + tzz); t(0, 1) = txy - twz;          t(0, 2) = txz + twy;
    t(1, 0) = txy + twz;          t(1, 1) = 1.0f - (txx + tzz); t(1, 2) = tyz - twx;
    t(2, 0) = txz - twy;          t(2, 1) = tyz + twx;          t(2, 2) = 1.0f - (txx + tyy);
}

// /root/reference/src/math_utils.h:6-10: re-orthonormalise the rotation through a unit quaternion
inline void Normalize(Mat4f& pose)
{
    SetRotation(pose, Normalized(QuatFromRotation(pose)));
}

// /root/reference/src/math_utils.h:12-25: Matrix4f -> [angle * axis, translation]
template <typename T>
inline void MatrixToPose(const Mat4f& mat, std::array<T, 6>& pose)
{
    const Quatf q = QuatFromRotation(mat);
    float n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    float angle = 0.0f, ax = 1.0f, ay = 0.0f, az = 0.0f;
    if (n != 0.0f) {
        angle = 2.0f * std::atan2(n, std::fabs(q.w));
        if (q.w < 0.0f) n = -n;
        ax = q.x / n; ay = q.y / n; az = q.z / n;
    }
    pose[0] = ax * angle; pose[1] = ay * angle; pose[2] = az * angle;
    pose[3] = mat(0, 3); pose[4] = mat(1, 3); pose[5] = mat(2, 3);
}

// /root/reference/src/math_utils.h:27-41: [r, t] -> Matrix4f.  A zero rotation vector stays a zero axis
// (Eigen's normalized() returns it unchanged), which yields the identity rotation.
template <typename T>
inline void PoseToMatrix(const std::array<T, 6>& pose, Mat4f& mat)
{
    mat = Mat4f::Identity();
    float r[3] = {(float)pose[0], (float)pose[1], (float)pose[2]};
    const float f = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (f > 0.0f) { r[0] /= f; r[1] /= f; r[2] /= f; }
    const float s = std::sin(f), c = std::cos(f);
    const float sx = s * r[0], sy = s * r[1], sz = s * r[2];
    const float cx = (1.0f - c) * r[0], cy = (1.0f - c) * r[1], cz = (1.0f - c) * r[2];
    float tmp;
    tmp = cx * r[1]; mat(0, 1) = tmp - sz; mat(1, 0) = tmp + sz;
    tmp = cx * r[2]; mat(0, 2) = tmp + sy; mat(2, 0) = tmp - sy;
    tmp = cy * r[2]; mat(1, 2) = tmp - sx; mat(2, 1) = tmp + sx;
    mat(0, 0) = cx * r[0] + c; mat(1, 1) = cy * r[1] + c; mat(2, 2) = cz * r[2] + c;
    mat(0, 3) = (float)pose[3]; mat(1, 3) = (float)pose[4]; mat(2, 3) = (float)pose[5];
}

}  // namespace soslam_host
