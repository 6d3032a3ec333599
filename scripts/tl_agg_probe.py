#!/usr/bin/env python3
"""Development probe: the two-level PCG's aggregate size (SOSLAM_TL_AGG, read at set_problem) on BASELINE configs[2] plus 10 % of
tracks of length 20: PCG iterations and time per LM iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stereo_orb_slam_amd import ba, synth


def merge(p, q):
    oc = np.concatenate([p.obs_cam, q.obs_cam]); op = np.concatenate([p.obs_pt, q.obs_pt + np.uint32(p.n_pt)])
    uv = np.concatenate([p.obs_uv, q.obs_uv]); order = np.lexsort((op, oc))
    return synth.BaProblem(p.poses_wc, np.concatenate([p.points, q.points]), oc[order], op[order], uv[order], p.proj_l, p.proj_r)


p = merge(synth.generate_ba(3), synth.generate_ba(None, n_cam=500, n_pt=10000, track_mode=0, track_len=20))
for agg in sys.argv[1:]:
    os.environ["SOSLAM_TL_AGG"] = agg
    with ba.BundleAdjustment(ba.default_options(linear_solver=2, pcg_tolerance=1e-8, pcg_max_iterations=2000)) as h:
        h.load(p)
        h.iterate(2)
        h.set_state(p.poses_cw(), p.points_f64())
        t0 = time.perf_counter(); s = h.iterate(10); dt = time.perf_counter() - t0
        log = h.iteration_log()
    print(f"aggregates of {agg}: {dt / 10 * 1e3:.3f} ms/iteration, PCG iterations {[it.linear_iterations for it in log[1:]]}, final cost {s.final_cost:.9e}", flush=True)
