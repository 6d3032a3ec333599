#!/usr/bin/env python3
"""Profiling driver: load BASELINE.json configs[config-1] and launch ONE kernel of the hot path `reps` times through
soslam_ba_time_kernel (for rocprofv3 --pmc / --kernel-trace runs).  usage: kernel_loop.py KERNEL [config] [reps]
KERNEL: linearize | cost | point_reduce | schur | backsub"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_orb_slam_amd import _lib, ba, synth

kid = {"linearize": _lib.KERNEL_LINEARIZE, "cost": _lib.KERNEL_COST, "point_reduce": _lib.KERNEL_POINT_REDUCE,
       "schur": _lib.KERNEL_SCHUR, "backsub": _lib.KERNEL_BACKSUB}[sys.argv[1]]
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
with ba.BundleAdjustment(ba.default_options(linear_solver=2)) as h:
    h.load(synth.generate_ba(cfg))
    print(sys.argv[1], "avg ms", h.time_kernel(kid, reps))
