"""Development probe: a chain of keyframes with a few loop closures (the reference's own pose graphs): the band factor of the odometry
chain as preconditioner against the two-level PCG, by number of closures."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereo_orb_slam_amd import pg, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for loops in (1, 2, 3, 4, 6, 12):
    c = synth.generate_pg(5, n_node=n, row_len=n, n_loop_max=loops, min_gap=50, radius=80.0)
    for pre, name in ((3, "band factor"), (2, "two-level")):
        with pg.PoseGraph(pg.default_options(max_iterations=10, preconditioner=pre)) as h:
            ts = []
            for _ in range(3):
                h.load(c)
                t0 = time.perf_counter(); s = h.optimize(); ts.append(time.perf_counter() - t0)
            per = [it.linear_iterations // max(1, it.trials) for it in h.iteration_log()]
        print(f"{n} vertices, {len(c.e_from) - n + 1} closures, {name}: {min(ts) * 1e3:.2f} ms, {s.linear_iterations} linear iterations {per}", flush=True)
