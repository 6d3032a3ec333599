// ba_demo.cpp - plays the part of /root/reference/src/slam.cpp around the back-end: load a map in the
// reference's Dump format, set the stereo projection, run BundleAdjuster::Optimize on a frame range (or the
// per-frame + sliding-window schedule of slam.cpp:121-129), write the result as a Dump.
//
//   ba_demo <in_folder> <out_folder> [--start S] [--end E] [--schedule INTERVAL] [--iters N] [--quiet] [--shard-one-rank]
//           [--proj fx cx cy tx]   (rectified rig: P_l = K[I|0], P_r = K[I|t], P_r[3] = tx; default KITTI-00)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bundle_adjuster.h"
#include "dump_io.h"
#include "reprojection_error.h"

static void InitializeStereoReprojectionError(float fx, float cx, float cy, float tx)
{
    // float32 camera parameters widened to double, as /root/reference/src/slam.cpp:176-209 does
    const std::array<double, 12> pl = {fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0};
    const std::array<double, 12> pr = {fx, 0, cx, tx, 0, fx, cy, 0, 0, 0, 1, 0};
    ReprojectionError::SetLeftProjection(pl);
    ReprojectionError::SetRightProjection(pr);
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <in_folder> <out_folder> [--start S] [--end E] [--schedule INTERVAL] [--iters N] [--quiet]\n", argv[0]);
        return 2;
    }
    long start = 0, end = -1, schedule = 0, iters = -1;
    bool quiet = false, shard_one_rank = false;
    float fx = 718.856f, cx = 607.1928f, cy = 185.2157f, tx = -386.1448f;
    for (int i = 3; i < argc; i++) {
        if (!std::strcmp(argv[i], "--start") && i + 1 < argc) start = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--end") && i + 1 < argc) end = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--schedule") && i + 1 < argc) schedule = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--iters") && i + 1 < argc) iters = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
        else if (!std::strcmp(argv[i], "--shard-one-rank")) shard_one_rank = true;
        else if (!std::strcmp(argv[i], "--proj") && i + 4 < argc) { fx = (float)std::atof(argv[i + 1]); cx = (float)std::atof(argv[i + 2]); cy = (float)std::atof(argv[i + 3]); tx = (float)std::atof(argv[i + 4]); i += 4; }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    soslam_host::MapState map;
    if (!soslam_host::ReadDump(argv[1], map)) {
        std::fprintf(stderr, "[FAIL]: cannot read a dump from %s\n", argv[1]);
        return 1;
    }
    InitializeStereoReprojectionError(fx, cx, cy, tx);
    BundleAdjuster ba(map.frames, map.points);
    if (quiet) ba.Options().verbose = 0;
    if (shard_one_rank) {
        // the multi-GPU path of the shim with the one rank a one-GPU box allows: unique id, communicator, sharding (the
        // whole window is this rank's share), job-wide pattern, all-reduces, global read-back
        unsigned char id[SOSLAM_RCCL_UNIQUE_ID_BYTES];
        if (soslam_rccl_get_unique_id(id) != SOSLAM_OK) {
            std::fprintf(stderr, "[FAIL]: no RCCL unique id: %s\n", soslam_last_error());
            return 1;
        }
        ba.EnableSharding(0, 1, id);
    }
    if (iters >= 0) ba.Options().max_iterations = (int32_t)iters;
    int rc = 0;
    if (schedule > 0) {
        // slam.cpp:121-129 - after every frame a structure-only pass on that frame, every `interval` frames a
        // window of 2 * interval frames
        for (long n = 1; n <= (long)map.frames.size(); n++) {
            ba.Optimize((unsigned)(n - 1), (unsigned)n);
            rc |= ba.LastStatus();
            if (n % schedule == 0) {
                const long s = n - 2 * schedule > 0 ? n - 2 * schedule : 0;
                ba.Optimize((unsigned)s, (unsigned)n);
                rc |= ba.LastStatus();
            }
        }
    } else {
        if (end < 0) end = (long)map.frames.size();
        ba.Optimize((unsigned)start, (unsigned)end);
        rc = ba.LastStatus();
        const soslam_ba_summary& s = ba.LastSummary();
        std::printf("RESULT status %d initial %.17g final %.17g iterations %d accepted %d termination %d\n", rc, s.initial_cost,
                    s.final_cost, s.iterations, s.accepted, s.termination);
    }
    if (!soslam_host::WriteDump(argv[2], map.frames, map.points)) {
        std::fprintf(stderr, "[FAIL]: cannot write the dump to %s\n", argv[2]);
        return 1;
    }
    return rc == 0 ? 0 : 3;
}
