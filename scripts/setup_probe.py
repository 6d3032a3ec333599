"""Where a kept handle's set_problem goes for the reference's small calls (SOSLAM_SETUP_TIMING=1 prints the marks)."""
import os
import sys
import time

os.environ["SOSLAM_SETUP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from stereo_orb_slam_amd import ba, synth

for name, kw in (("frame", dict(n_cam=1, n_pt=1000, track_mode=0, track_len=1)), ("window", dict(n_cam=20, n_pt=6000, track_mode=1, track_len=6))):
    p = synth.generate_ba(None, **kw)
    o = ba.default_options(max_iterations=10)
    h = ba.BundleAdjustment(o)
    for i in range(3):
        print(f"--- {name} load {i}", flush=True)
        t0 = time.perf_counter()
        h.load(p)
        t1 = time.perf_counter()
        s = h.solve()
        t2 = time.perf_counter()
        h.get_state()
        t3 = time.perf_counter()
        print(f"load {1e3 * (t1 - t0):.3f} ms (set-up inside {1e3 * s.setup_seconds:.3f}), solve {1e3 * (t2 - t1):.3f} (inside {1e3 * s.solve_seconds:.3f}), "
              f"get_state {1e3 * (t3 - t2):.3f}", flush=True)
    h.close()
