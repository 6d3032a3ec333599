#!/usr/bin/env python3
"""Development probe: PCG iterations and time per LM iteration (two-level PCG against block-Jacobi PCG) on problems whose reduced
matrix is wider than the cyclic reduction's band: (a) tracks of 24 cameras on a 150-camera chain, (b) BASELINE configs[2] plus 10 % of tracks of length 20."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from stereo_orb_slam_amd import ba, synth


def merge(p, q):
    oc = np.concatenate([p.obs_cam, q.obs_cam]); op = np.concatenate([p.obs_pt, q.obs_pt + np.uint32(p.n_pt)])
    uv = np.concatenate([p.obs_uv, q.obs_uv]); order = np.lexsort((op, oc))
    return synth.BaProblem(p.poses_wc, np.concatenate([p.points, q.points]), oc[order], op[order], uv[order], p.proj_l, p.proj_r)


probs = {"chain150_track24": synth.generate_ba(None, n_cam=150, n_pt=6000, track_mode=0, track_len=24, spacing=0.3)}
if len(sys.argv) > 1:
    probs["config3_plus_10pct_len20"] = merge(synth.generate_ba(3), synth.generate_ba(None, n_cam=500, n_pt=10000, track_mode=0, track_len=20))
for name, p in probs.items():
    for env in ("", "1"):
        if env:
            os.environ["SOSLAM_NO_TWO_LEVEL"] = "1"
        else:
            os.environ.pop("SOSLAM_NO_TWO_LEVEL", None)
        with ba.BundleAdjustment(ba.default_options(linear_solver=2, pcg_tolerance=1e-8, pcg_max_iterations=2000)) as h:
            h.load(p)
            h.iterate(2)
            h.set_state(p.poses_cw(), p.points_f64())
            t0 = time.perf_counter(); s = h.iterate(10); dt = time.perf_counter() - t0
            log = h.iteration_log()
        print(f"{name} ({p.n_cam} cams, {p.n_obs} obs) {'block-Jacobi PCG' if env else 'two-level PCG'}: {dt / 10 * 1e3:.3f} ms/iteration, "
              f"PCG rounds {[it.linear_iterations for it in log[1:]]}, final cost {s.final_cost:.9e}", flush=True)

# parity of the two paths and the oracle on a smaller problem of the second kind
import oracle
small = merge(synth.generate_ba(None, n_cam=120, n_pt=24000, track_mode=0, track_len=10), synth.generate_ba(None, n_cam=120, n_pt=2400, track_mode=0, track_len=20))
res = {}
for env in ("", "1"):
    if env:
        os.environ["SOSLAM_NO_TWO_LEVEL"] = "1"
    else:
        os.environ.pop("SOSLAM_NO_TWO_LEVEL", None)
    with ba.BundleAdjustment(ba.default_options(linear_solver=2, pcg_tolerance=1e-10, pcg_max_iterations=4000)) as h:
        h.load(small)
        h.iterate(12)
        res[env] = ([it.cost for it in h.iteration_log()], [it.linear_iterations for it in h.iteration_log()], h.get_state()[0])
o = oracle.default_options(max_iterations=12, check_termination=0, num_threads=8)
oc, op_, osum, olog = oracle.solve(small.obs_cam, small.obs_pt, small.obs_uv, small.poses_cw(), small.points_f64(), small.proj_l, small.proj_r, small.cam_fixed, o)
oc_cost = [e.cost for e in olog]
for env in ("", "1"):
    c, li, cams = res[env]
    print("two-level" if not env else "blockjacobi", "rounds", li[1:], "max rel cost diff vs oracle", max(abs(a - b) / b for a, b in zip(c, oc_cost)), "max pose diff", float(np.abs(cams - oc).max()))

if len(sys.argv) > 1:
    big = probs["config3_plus_10pct_len20"]
    o = oracle.default_options(max_iterations=12, check_termination=0, num_threads=16)
    # the probe above: 2 warm-up iterations, reset, 10 iterations -> compare a plain 10-iteration run
    o.max_iterations = 10
    t0 = time.perf_counter()
    _, _, osum, olog = oracle.solve(big.obs_cam, big.obs_pt, big.obs_uv, big.poses_cw(), big.points_f64(), big.proj_l, big.proj_r, big.cam_fixed, o)
    print(f"oracle on config3_plus_10pct_len20: final cost {osum.final_cost:.9e} after 10 iterations ({time.perf_counter() - t0:.1f} s); costs {[f'{e.cost:.6e}' for e in olog]}")
