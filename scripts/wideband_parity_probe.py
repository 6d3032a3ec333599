#!/usr/bin/env python3
"""Development probe: LM trajectories of a wide-band problem (configs[2] plus 10 % of tracks of length 20) under the two-level and the
block-Jacobi PCG at several tolerances against the oracle's (direct solve)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from stereo_orb_slam_amd import ba, synth
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def merge(p, q):
    oc = np.concatenate([p.obs_cam, q.obs_cam]); op = np.concatenate([p.obs_pt, q.obs_pt + np.uint32(p.n_pt)])
    uv = np.concatenate([p.obs_uv, q.obs_uv]); order = np.lexsort((op, oc))
    return synth.BaProblem(p.poses_wc, np.concatenate([p.points, q.points]), oc[order], op[order], uv[order], p.proj_l, p.proj_r)


big = merge(synth.generate_ba(3), synth.generate_ba(None, n_cam=500, n_pt=10000, track_mode=0, track_len=20))
o = oracle.default_options(max_iterations=6, check_termination=0, num_threads=16)
_, _, osum, olog = oracle.solve(big.obs_cam, big.obs_pt, big.obs_uv, big.poses_cw(), big.points_f64(), big.proj_l, big.proj_r, big.cam_fixed, o)
print("oracle     ", [f"{e.cost:.6e}" for e in olog])
print("oracle it 1: model", olog[1].model_cost_change, "cand", olog[1].candidate_cost, "step", olog[1].step_norm, "rho", olog[1].relative_decrease, "radius", olog[1].radius)
with ba.BundleAdjustment(ba.default_options(linear_solver=2, pcg_tolerance=1e-10, pcg_max_iterations=4000, lower_bound=-float("inf"), upper_bound=float("inf"))) as h:
    h.load(big)
    s_ = h.iterate(3)
    log = h.iteration_log()
print("unbounded   ", [f"{it.cost:.6e}" for it in log], "it 1: model", log[1].model_cost_change, "cand", log[1].candidate_cost, "step", log[1].step_norm, "rho", log[1].relative_decrease)
oo = oracle.default_options(max_iterations=3, check_termination=0, num_threads=16, lower_bound=-float("inf"), upper_bound=float("inf"))
_, _, _, ol2 = oracle.solve(big.obs_cam, big.obs_pt, big.obs_uv, big.poses_cw(), big.points_f64(), big.proj_l, big.proj_r, big.cam_fixed, oo)
print("oracle unbounded", [f"{e.cost:.6e}" for e in ol2], "it 1: cand", ol2[1].candidate_cost, "step", ol2[1].step_norm)
