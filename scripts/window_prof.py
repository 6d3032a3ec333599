"""The 20-frame window call of the reference's schedule (slam.cpp:126-129), repeated: the workload of a kernel-trace profile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_orb_slam_amd import ba, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
# second argument "frame": the per-frame call (slam.cpp:123: one constant frame, its points) instead of the window
if len(sys.argv) > 2 and sys.argv[2] == "frame":
    p = synth.generate_ba(None, n_cam=1, n_pt=1000, track_mode=0, track_len=1)
else:
    p = synth.generate_ba(None, n_cam=20, n_pt=6000, track_mode=1, track_len=6)
h = ba.BundleAdjustment(ba.default_options(max_iterations=10))
for _ in range(reps):
    h.load(p)
    s = h.solve()
    h.get_state()
print(f"{p.n_obs} observations, {s.iterations} iterations, solve {1e3 * s.solve_seconds:.3f} ms, set-up {1e3 * s.setup_seconds:.3f} ms")
h.close()
