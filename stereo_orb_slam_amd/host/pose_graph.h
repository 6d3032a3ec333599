// /root/reference/src/pose_graph.h:1-3: a loop edge is a pair of frame ids (first < second by construction).
#pragma once
#include <utility>
typedef std::pair<int, int> PoseGraphEdge;
