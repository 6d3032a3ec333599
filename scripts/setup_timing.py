"""Development aid: host phases of soslam_ba_set_problem at a BASELINE.json config (SOSLAM_SETUP_TIMING=1 prints them) with a
kept handle.  usage: SOSLAM_SETUP_TIMING=1 python scripts/setup_timing.py [config]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_orb_slam_amd import ba, synth

p = synth.generate_ba(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
h = ba.BundleAdjustment(ba.default_options(linear_solver=2))
h.load(p)
t0 = time.perf_counter()
h.load(p)
print("second load (kept handle) ms", (time.perf_counter() - t0) * 1e3)
