#!/usr/bin/env python3
"""Development probe: LM iterations of a synthetic config with the per-iteration log (PCG rounds, residual) printed.
Usage: python scripts/lag_probe.py [config] [iterations] [pcg_tol]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereo_orb_slam_amd import ba, synth

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8
prob = synth.generate_ba(cfg)
opts = ba.default_options(device=0, linear_solver=2, pcg_tolerance=tol, verbose=1)
with ba.BundleAdjustment(opts) as h:
    h.load(prob)
    h.iterate(3)
    h.set_state(prob.poses_cw(), prob.points_f64())
    t0 = time.perf_counter()
    s = h.iterate(iters)
    dt = time.perf_counter() - t0
    d = ba.summary_dict(s)
    print(f"RESULT cfg {cfg} iters {iters} tol {tol}: {iters / dt:.1f} it/s, final cost {d['final_cost']:.12e}, linear iterations {d['linear_iterations']}")
