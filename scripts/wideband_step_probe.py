#!/usr/bin/env python3
"""Development probe: one LM step of the wide-band problem (configs[2] plus 10 % of tracks of length 20) on the device against the
oracle, point by point."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from stereo_orb_slam_amd import _lib as L, ba, synth


def merge(p, q):
    oc = np.concatenate([p.obs_cam, q.obs_cam]); op = np.concatenate([p.obs_pt, q.obs_pt + np.uint32(p.n_pt)])
    uv = np.concatenate([p.obs_uv, q.obs_uv]); order = np.lexsort((op, oc))
    return synth.BaProblem(p.poses_wc, np.concatenate([p.points, q.points]), oc[order], op[order], uv[order], p.proj_l, p.proj_r)


n_cam = int(sys.argv[1]) if len(sys.argv) > 1 else 500
scale = n_cam / 500.0
big = merge(synth.generate_ba(None, n_cam=n_cam, n_pt=int(100000 * scale), track_mode=0, track_len=10),
            synth.generate_ba(None, n_cam=n_cam, n_pt=int(10000 * scale), track_mode=0, track_len=20))
ref = oracle.step(big.obs_cam, big.obs_pt, big.obs_uv, big.poses_cw(), big.points_f64(), big.proj_l, big.proj_r, big.cam_fixed, 1e4)
with ba.BundleAdjustment(ba.default_options(linear_solver=2, pcg_tolerance=1e-12, pcg_max_iterations=4000)) as h:
    h.load(big)
    h.debug_step(1e4)
    dc, dp, sc = h.debug_read(L.DBG_STEP_CAM), h.debug_read(L.DBG_STEP_POINT), h.debug_read(L.DBG_STEP_SCALARS)
print("scalars", sc)
print("dc: max |gpu - oracle|", np.abs(dc - ref["dc"]).max(), "max |oracle|", np.abs(ref["dc"]).max())
err = np.abs(dp - ref["dp"]).max(axis=1)
print("dp: max |gpu - oracle|", err.max(), "max |oracle dp|", np.abs(ref["dp"]).max(), "max |gpu dp|", np.abs(dp).max())
bad = np.argsort(err)[-8:][::-1]
cnt = np.bincount(big.obs_pt, minlength=big.n_pt)
for p in bad:
    cams = np.sort(big.obs_cam[big.obs_pt == p])
    print(f"point {p}: track {cnt[p]} cameras {cams.min()}..{cams.max()}  gpu dp {dp[p]}  oracle dp {ref['dp'][p]}")
print("points with error > 1e-6 * max|dp|:", int((err > 1e-6 * np.abs(ref['dp']).max()).sum()), "of", big.n_pt, "; of them with track 20:", int(((err > 1e-6 * np.abs(ref['dp']).max()) & (cnt == 20)).sum()))
