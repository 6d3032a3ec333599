#!/usr/bin/env python3
"""Per-kernel HIP-event timings of the BA hot path at a synthetic config (development aid).
usage: python scripts/kernel_times.py [config=3] [reps=30]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_orb_slam_amd import ba, synth  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
p = synth.generate_ba(cfg)
h = ba.BundleAdjustment(ba.default_options(linear_solver=3))
h.load(p)
tag = os.environ.get("SOSLAM_LIN_VARIANT", "")
for k, name, bpo in ((0, "ba_linearize", 128), (1, "ba_cost", 48), (2, "ba_point_reduce", 88), (3, "ba_schur", 56), (4, "ba_backsub", 56)):
    ms = h.time_kernel(k, reps)
    print(f"{tag} {name:16s} {ms * 1e3:9.1f} us  {bpo * p.n_obs / ms / 1e6:8.1f} GB/s")
