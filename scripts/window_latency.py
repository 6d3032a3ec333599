"""Latency of the reference's BA schedule (slam.cpp:121-129): a per-frame call (one fixed camera, structure only) and a
sliding-window call (2 x refine_interval = 20 frames), one-shot handle versus a handle kept across calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from stereo_orb_slam_amd import ba, synth


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


for name, kw, fixed_all in (("per-frame (1 frame, ~1000 points)", dict(n_cam=1, n_pt=1000, track_mode=0, track_len=1), True),
                            ("sliding window (20 frames, 6000 points)", dict(n_cam=20, n_pt=6000, track_mode=1, track_len=6), False)):
    p = synth.generate_ba(None, **kw)
    o = ba.default_options(max_iterations=10)
    one = timed(lambda: ba.optimize(p, o), 20)
    h = ba.BundleAdjustment(o)

    p0, x0 = p.poses_cw(), p.points_f64()   # the C ABI's formats, converted once (the host shim's conversion is not what is measured)

    def kept():
        h.load(p, p0, x0)
        s = h.solve()
        h.get_state()
        return s
    keep = timed(kept, 20)
    s = kept()
    h.close()
    print(f"{name}: {p.n_obs} obs: one-shot {one:.2f} ms/call, kept handle {keep:.2f} ms/call "
          f"(index build + upload {1e3 * s.setup_seconds:.2f} ms, {s.iterations} iterations {1e3 * s.solve_seconds:.2f} ms)", flush=True)
