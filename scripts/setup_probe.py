import sys
sys.path.insert(0, "/root/repo")
from stereo_orb_slam_amd import ba, synth
p = synth.generate_ba(None, n_cam=20, n_pt=6000, track_mode=1, track_len=6)
with ba.BundleAdjustment(ba.default_options(max_iterations=2)) as h:
    h.load(p); h.solve()
    print("---- second load", file=sys.stderr, flush=True)
    h.load(p)
