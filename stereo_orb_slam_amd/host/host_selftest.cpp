// host_selftest.cpp - CPU-only checks of the host shim's container semantics (no GPU, no solver call):
// the pose-setter side effects of camera_frame.h and the float32 conversions of mat4f.h.
// Prints "OK" and returns 0 when every check holds.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "camera_frame.h"
#include "dump_io.h"
#include "mat4f.h"

using soslam_host::Mat4f;

static int g_fail = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
    } while (0)

static Mat4f MakePose(double rx, double ry, double rz, double tx, double ty, double tz)
{
    Mat4f m;
    soslam_host::PoseToMatrix(std::array<double, 6>{rx, ry, rz, tx, ty, tz}, m);
    return m;
}

static float MaxDiff(const Mat4f& a, const Mat4f& b)
{
    float d = 0;
    for (int i = 0; i < 16; i++) d = std::fmax(d, std::fabs(a.m[i] - b.m[i]));
    return d;
}

int main(int argc, char** argv)
{
    // MatrixToPose(PoseToMatrix(p)) == p, rotation stays orthonormal
    const std::array<double, 6> p = {0.1, -0.25, 0.4, 1.0, -2.0, 3.0};
    Mat4f m;
    soslam_host::PoseToMatrix(p, m);
    std::array<double, 6> q;
    soslam_host::MatrixToPose(m, q);
    for (int i = 0; i < 6; i++) CHECK(std::fabs(q[i] - p[i]) < 2e-6);
    CHECK(MaxDiff(m * m.inverse(), Mat4f::Identity()) < 1e-5f);
    // zero rotation vector -> identity rotation (Eigen leaves a zero axis unchanged)
    soslam_host::PoseToMatrix(std::array<double, 6>{0, 0, 0, 1, 2, 3}, m);
    CHECK(m(0, 0) == 1.0f && m(1, 1) == 1.0f && m(2, 2) == 1.0f && m(0, 1) == 0.0f && m(0, 3) == 1.0f);
    // rotation by pi about x (trace < 0 branch of the quaternion conversion)
    soslam_host::PoseToMatrix(std::array<double, 6>{3.14159265, 0, 0, 0, 0, 0}, m);
    soslam_host::MatrixToPose(m, q);
    CHECK(std::fabs(std::fabs(q[0]) - 3.14159265) < 1e-3 && std::fabs(q[1]) < 1e-3 && std::fabs(q[2]) < 1e-3);

    // frame chain: global = previous global * relative
    const Mat4f r0 = MakePose(0, 0, 0, 0, 0, 0), r1 = MakePose(0, 0.1, 0, 0, 0, 1), r2 = MakePose(0.05, 0, 0, 0.2, 0, 1);
    Frame f0(nullptr, r0), f1(&f0, r1), f2(&f1, r2);
    CHECK(MaxDiff(f2.GlobalPose(), r1 * r2) < 1e-6f);

    // GlobalPose(T) moves points FIRST observed in that frame by T * old^-1, others stay
    MapPoint a(1, 2, 10), b(-1, 0, 5);
    f1.AddMapPoints(&a, true);
    f1.AddMapPoints(&b, false);
    const Mat4f old1 = f1.GlobalPose();
    const Mat4f new1 = MakePose(0, 0.12, 0.01, 0.05, 0, 1.1);
    MapPoint a_expect = a;
    a_expect.Transform(new1 * old1.inverse());
    f1.GlobalPose(new1);
    CHECK(std::fabs(a.Position()[0] - a_expect.Position()[0]) < 1e-5f && std::fabs(a.Position()[2] - a_expect.Position()[2]) < 1e-5f);
    CHECK(b.Position()[0] == -1.0f && b.Position()[2] == 5.0f);
    CHECK(MaxDiff(f1.GlobalPose(), new1) < 1e-6f);
    // relative pose re-derived against the previous frame's current global pose
    CHECK(MaxDiff(f0.GlobalPose() * f1.RelativePose(), f1.GlobalPose()) < 1e-6f);
    // the next frame keeps its OLD global pose until UpdatePose() re-chains it
    CHECK(MaxDiff(f2.GlobalPose(), r1 * r2) < 1e-6f);
    f2.UpdatePose();
    CHECK(MaxDiff(f2.GlobalPose(), f1.GlobalPose() * r2) < 1e-5f);

    // MapPoint::Position(double[3]) narrows to float32
    MapPoint c(0, 0, 0);
    c.Position(std::array<double, 3>{1.0000000001, 2.5, -3.25});
    CHECK(c.Position()[0] == 1.0f && c.Position()[1] == 2.5f && c.Position()[2] == -3.25f);

    // Dump round trip (only if a scratch folder is given)
    if (argc > 1) {
        f0.AddObservation(Observation(0, 10.5f, 20.25f, 8.5f, 20.25f, 1.0f));
        f0.AddMapPoints(&a, true);
        f1.AddObservation(Observation(0, 11.5f, 21.25f, 9.5f, 21.25f, 1.0f));
        f1.AddObservation(Observation(1, 300.0f, 100.0f, 280.0f, 100.0f, 1.0f));
        std::vector<Frame*> frames = {&f0, &f1, &f2};
        std::vector<MapPoint*> points = {&a, &b};
        CHECK(soslam_host::WriteDump(argv[1], frames, points));
        soslam_host::MapState back;
        CHECK(soslam_host::ReadDump(argv[1], back));
        CHECK(back.frames.size() == 3 && back.points.size() == 2);
        if (back.frames.size() == 3) {
            for (int i = 0; i < 3; i++) CHECK(MaxDiff(back.frames[i]->GlobalPose(), frames[i]->GlobalPose()) < 2e-6f);
            CHECK(back.frames[1]->Observations().size() == 2);
            CHECK(back.frames[1]->Observations()[1].point_id == 1 && back.frames[1]->Observations()[1].u_r == 280.0f);
            CHECK(back.points[0]->Position() == a.Position());
        }
        soslam_host::MapState missing;
        CHECK(!soslam_host::ReadDump(std::string(argv[1]) + "/does_not_exist", missing));
    }
    if (g_fail == 0) std::printf("OK\n");
    return g_fail == 0 ? 0 : 1;
}
