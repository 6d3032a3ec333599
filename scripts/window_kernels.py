"""Development aid: the sliding-window call of scripts/window_latency.py (20 frames, 27 k observations, 10 iterations) repeated
with a kept handle, for a rocprofv3 --kernel-trace --stats run."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_orb_slam_amd import ba, synth

p = synth.generate_ba(None, n_cam=20, n_pt=6000, track_mode=1, track_len=6)
h = ba.BundleAdjustment(ba.default_options(max_iterations=10))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    h.load(p)
    s = h.solve()
print(s.iterations, s.final_cost)
