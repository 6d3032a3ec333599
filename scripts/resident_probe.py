"""The resident structure-only solve (ba_points_solve: controller and line search on the device) against the host-driven loop
(SOSLAM_NO_RESIDENT_SOLVE=1): iteration logs and final points of the per-frame call, and the time of a call with the handle kept.
Usage: resident_probe.py out.npz   (run once with and once without the switch, then resident_probe.py --compare a.npz b.npz)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    worst = 0.0
    for k in a.files:
        x, y = a[k], b[k]
        if x.shape != y.shape:
            print(f"{k}: shapes {x.shape} {y.shape}")
            worst = 1.0
            continue
        d = float(np.max(np.abs(x - y) / (1e-300 + np.maximum(np.abs(x), np.abs(y))))) if x.size else 0.0
        worst = max(worst, d)
        print(f"{k}: shape {x.shape} worst relative difference {d:.3e}")
    print("WORST", worst)
    sys.exit(0)

from stereo_orb_slam_amd import ba, synth

out = {}
cases = (("frame", dict(n_cam=1, n_pt=1000, track_mode=0, track_len=1), {}),
         ("frame_tight", dict(n_cam=1, n_pt=1000, track_mode=0, track_len=1), dict(lower_bound=-3.0, upper_bound=3.0)),
         ("frame400", dict(n_cam=1, n_pt=400, track_mode=0, track_len=1), {}),
         ("three", dict(n_cam=3, n_pt=3000, track_mode=1, track_len=3), {}),
         ("unbounded", dict(n_cam=1, n_pt=1000, track_mode=0, track_len=1), dict(lower_bound=-np.inf, upper_bound=np.inf)))
for name, kw, okw in cases:
    p = synth.generate_ba(None, **kw)
    if kw["n_cam"] > 1:
        p.cam_fixed[:] = 1
    o = ba.default_options(max_iterations=10, **okw)
    h = ba.BundleAdjustment(o)

    def call():
        h.load(p)
        s = h.solve()
        st = h.get_state()
        return s, st
    s, (poses, pts) = call()
    log = h.iteration_log()
    out[name + "_pts"] = pts
    out[name + "_log"] = np.array([[e.cost, e.candidate_cost, e.model_cost_change, e.relative_decrease, e.radius, e.step_norm,
                                    e.gradient_max_norm, e.accepted, e.valid] for e in log], dtype=np.float64)
    out[name + "_sum"] = np.array([s.initial_cost, s.final_cost, s.iterations, s.accepted, s.termination, s.line_search_steps], dtype=np.float64)
    for _ in range(5):
        call()
    t0 = time.perf_counter()
    for _ in range(50):
        call()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50):
        h.solve()
    ms_solve = (time.perf_counter() - t0) / 50 * 1e3
    print(f"{name}: {p.n_obs} obs, {s.iterations} iterations, {s.line_search_steps} line-search steps, termination {s.termination}, "
          f"cost {s.initial_cost:.6e} -> {s.final_cost:.6e}; call {ms:.3f} ms (solve {1e3 * s.solve_seconds:.3f}, set-up {1e3 * s.setup_seconds:.3f}); "
          f"repeat solve at the minimum {ms_solve:.3f} ms", flush=True)
    h.close()
np.savez(sys.argv[1], **out)
