"""Pose-graph path timings (development aid): PoseGraphOptimizer::Optimize-sized solves on the GPU against the CPU oracle.
usage: python scripts/pg_times.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from stereo_orb_slam_amd import pg, synth

for cfg in (6, 5):
    g = synth.generate_pg(cfg)
    o = pg.default_options()
    with pg.PoseGraph(o) as h:
        h.load(g)
        t0 = time.perf_counter()
        s = h.optimize()
        dt = time.perf_counter() - t0
        lin_ms = h.time_linearize(20)
    import ctypes as C
    OL = oracle.lib()
    oo = oracle.PgOptions()
    OL.oracle_pg_options_default(C.byref(oo))
    est = g.est.copy()
    osum = oracle.PgSummary()
    log = (oracle.PgIteration * 64)()
    t0 = time.perf_counter()
    OL.oracle_pg_solve(len(est), len(g.e_from), est, g.fixed, g.e_from, g.e_to, np.ascontiguousarray(g.meas),
                       np.ascontiguousarray(g.info), C.byref(oo), C.byref(osum), C.cast(log, C.c_void_p))
    dt_cpu = time.perf_counter() - t0
    print(f"pg config {cfg}: {len(g.est)} vertices / {len(g.e_from)} edges: GPU {s.iterations} iterations in {dt * 1e3:.1f} ms "
          f"(chi2 {s.initial_chi2:.6e} -> {s.final_chi2:.6e}, linear iterations {s.linear_iterations}), linearise kernel {lin_ms * 1e3:.1f} us; "
          f"CPU oracle {dt_cpu * 1e3:.1f} ms (chi2 {osum.final_chi2:.6e})", flush=True)
