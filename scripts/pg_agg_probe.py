"""Development probe: the pose graph's aggregate target size (SOSLAM_PG_AGG, read when the graph is loaded) on configs[4]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_orb_slam_amd import pg, synth

g = synth.generate_pg(5)
for agg in sys.argv[1:]:
    os.environ["SOSLAM_PG_AGG"] = agg
    with pg.PoseGraph(pg.default_options()) as h:
        ts = []
        for _ in range(4):
            h.load(g)
            t0 = time.perf_counter(); s = h.optimize(); ts.append(time.perf_counter() - t0)
    print(f"target {agg}: {min(ts[1:]) * 1e3:.2f} ms, {s.linear_iterations} PCG iterations, chi2 {s.final_chi2:.9e}", flush=True)
