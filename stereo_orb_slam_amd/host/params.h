// params.h - the optimisation constants of /root/reference/src/params.h:29-47 (only the BA block: the visual
// odometry and loop-closure constants belong to components that are out of scope).
#pragma once

const unsigned int BA_MAX_ITERATION = 50;          // :34
const unsigned int BA_NUM_THREADS = 4;             // :37 - CPU threads of the reference; unused on the GPU path
// :40-41  the reference caps a solve at 1.0 s of wall clock (1e32 is its commented-out alternative).  The
// iterate then depends on the machine; BundleAdjuster keeps the cap as an option and defaults to OFF.
const float BA_MAX_TIME_SEC = 1e0;
const float BA_POINT_COORD_LOWER_BOUND = -10000.0; // :44
const float BA_POINT_COORD_UPPER_BOUND = 10000.0;  // :47
const unsigned int PG_NUM_ITERATIONS = 10;         // /root/reference/src/pose_graph_optimizer.cpp:69
