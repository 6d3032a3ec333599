"""Development aid: run the same step twice on fresh handles and report which read-back differs bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stereo_orb_slam_amd import ba, synth, _lib as L
for kw in (dict(n_cam=20, n_pt=3000, track_mode=1, track_len=6), dict(n_cam=1, n_pt=300, track_mode=0, track_len=1), dict(config=1)):
    cfg = kw.pop("config", None)
    p = synth.generate_ba(cfg, **kw)
    outs = []
    for rep in range(3):
        with ba.BundleAdjustment(ba.default_options(max_iterations=6)) as h:
            h.load(p)
            h.debug_step(1e4)
            o = {k: h.debug_read(getattr(L, "DBG_" + k)) for k in ("RESIDUALS", "JAC_POINT", "S_DENSE", "RHS", "STEP_CAM", "STEP_POINT", "STEP_SCALARS")}
            s = h.solve()
            o["final_cost"] = np.array([s.final_cost])
            outs.append(o)
    for k in outs[0]:
        same = all(np.array_equal(outs[0][k], o[k]) for o in outs[1:])
        print(p.name, k, "identical" if same else f"DIFFERS max {max(np.abs(outs[0][k] - o[k]).max() for o in outs[1:]):.3e}", flush=True)
