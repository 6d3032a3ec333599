import os, sys
sys.path.insert(0, "/root/repo")
from stereo_orb_slam_amd import ba, synth
for kw in (dict(n_cam=20, n_pt=6000, track_mode=1, track_len=6), dict(n_cam=20, n_pt=3000, track_mode=0, track_len=18)):
    p = synth.generate_ba(None, **kw)
    o = ba.default_options(max_iterations=10, profile_stages=1)
    with ba.BundleAdjustment(o) as h:
        h.load(p); h.solve(); h.load(p)
        s = h.solve()
        d = ba.summary_dict(s)
        print(kw, p.n_obs, "obs; solver", s.linear_solver, "iters", s.iterations, "solve ms", round(1e3*s.solve_seconds,2), {k: round(v/ max(1,s.iterations),4) for k,v in d["stage_ms"].items()})
