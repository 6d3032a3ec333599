// pg_demo.cpp - drives PoseGraphOptimizer::Optimize() the way slam.cpp:153 / loop_detector.cpp:146 do.
//   pg_demo <in_dump_folder> <out_dump_folder> [--loops loops.txt] [--skip-ba] [--graph out_graph.txt] [--quiet]
// loops.txt: one loop edge per line, "id_1 id_2" followed by the 16 floats (row-major 4x4) of its measured transform.
#include <cstdio>
#include <cstring>
#include <fstream>

#include "bundle_adjuster.h"
#include "dump_io.h"
#include "pose_graph_optimizer.h"
#include "reprojection_error.h"

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <in_folder> <out_folder> [--loops file] [--skip-ba] [--graph file] [--quiet]\n", argv[0]); return 2; }
    const char* loops = nullptr; const char* graph = nullptr;
    bool skip_ba = false, quiet = false;
    for (int i = 3; i < argc; i++) {
        if (!std::strcmp(argv[i], "--loops") && i + 1 < argc) loops = argv[++i];
        else if (!std::strcmp(argv[i], "--graph") && i + 1 < argc) graph = argv[++i];
        else if (!std::strcmp(argv[i], "--skip-ba")) skip_ba = true;
        else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    soslam_host::MapState map;
    if (!soslam_host::ReadDump(argv[1], map)) { std::fprintf(stderr, "[FAIL]: cannot read a dump from %s\n", argv[1]); return 1; }
    const float fx = 718.856f, cx = 607.1928f, cy = 185.2157f, tx = -386.1448f;
    ReprojectionError::SetLeftProjection({fx, 0, cx, 0, 0, fx, cy, 0, 0, 0, 1, 0});
    ReprojectionError::SetRightProjection({fx, 0, cx, tx, 0, fx, cy, 0, 0, 0, 1, 0});
    BundleAdjuster ba(map.frames, map.points);
    std::vector<PoseGraphEdge> edges;
    PoseGraphOptimizer po(ba, map.frames, edges);
    if (quiet) { ba.Options().verbose = 0; po.Options().verbose = 0; }
    po.RunGlobalBA(!skip_ba);
    if (loops) {
        std::ifstream f(loops);
        int a, b;
        while (f >> a >> b) {
            soslam_host::Mat4f t;
            for (int e = 0; e < 16; e++) f >> t.m[e];
            edges.emplace_back(a, b);
            po.AddLoopMeasurement(a, b, t);
        }
    }
    po.Optimize();
    const soslam_pg_summary& s = po.LastSummary();
    std::printf("RESULT status %d initial %.17g final %.17g iterations %d termination %d\n", po.LastStatus(), s.initial_chi2,
                s.final_chi2, s.iterations, s.termination);
    if (graph && !po.SavePoseGraph(graph)) return 1;
    if (!soslam_host::WriteDump(argv[2], map.frames, map.points)) return 1;
    return po.LastStatus() == 0 ? 0 : 3;
}
