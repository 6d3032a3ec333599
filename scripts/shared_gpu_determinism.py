"""Development aid: is one LM step bitwise reproducible while ANOTHER process uses the same GPU?
usage: python scripts/shared_gpu_determinism.py noise SECONDS   |   python scripts/shared_gpu_determinism.py check REPS [solver]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from stereo_orb_slam_amd import _lib as L
from stereo_orb_slam_amd import ba, synth

mode = sys.argv[1]
full = synth.generate_ba(3)
solver = int(sys.argv[3]) if len(sys.argv) > 3 else 2
h = ba.BundleAdjustment(ba.default_options(device=0, linear_solver=solver))
h.load(full)
if mode == "noise":
    t_end = time.time() + float(sys.argv[2])
    n = 0
    while time.time() < t_end:
        h.set_state(full.poses_cw(), full.points_f64())
        h.iterate(10)
        n += 10
    print("noise: iterations", n, flush=True)
else:
    reps = int(sys.argv[2])
    names = ["rows", "cost", "S", "rhs", "dc", "dp", "scalars"]
    what = [L.DBG_COMPACT_ROWS, L.DBG_COST, L.DBG_S_DENSE, L.DBG_RHS, L.DBG_STEP_CAM, L.DBG_STEP_POINT, L.DBG_STEP_SCALARS]
    ref = None
    diffs = {n: 0 for n in names}
    for r in range(reps):
        h.debug_step(1e4)
        cur = [h.debug_read(w).copy() for w in what]
        if ref is None:
            ref = cur
            continue
        for n, a, b in zip(names, ref, cur):
            if not np.array_equal(a, b):
                diffs[n] += 1
                if diffs[n] == 1:
                    d = np.abs(a - b)
                    print(f"rep {r}: {n} differs in {int((a != b).sum())} of {a.size} values, max abs {d.max():.3e}, max rel {np.nanmax(d / (np.abs(a) + 1e-300)):.3e}", flush=True)
    print("differing reps per output:", diffs, flush=True)
