"""Development aid: ONE process, the host-collective leg with an identity reduction (world 1), before and after the process
has created torch side streams and run a gloo collective on a device tensor.  Every run must repeat the first trajectory
bitwise.  usage: python scripts/single_proc_stage_probe.py [config] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stereo_orb_slam_amd import ba, synth

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
full = synth.generate_ba(cfg)
h = ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2))
h.load(full)
h.set_host_allreduce(lambda a, op: 0, 0, 1)
poses0, points0 = full.poses_cw(), full.points_f64()


def run(tag):
    h.set_state(poses0, points0)
    h.iterate(iters)
    log = h.iteration_log()
    d = np.array([[it.cost, it.candidate_cost, it.model_cost_change, it.valid, it.accepted, it.linear_iterations] for it in log])
    print(tag, "final cost %.12e" % d[-1, 0], "valid", int(d[1:, 3].sum()), "of", len(d) - 1, flush=True)
    return d


ref = run("run 0 (fresh process)")
for k in range(2):
    assert np.array_equal(run(f"run {k + 1} (no torch traffic)"), ref)
# torch side streams with a little work on each, as a process group's stream pool would be used
streams = [torch.cuda.Stream(dev) for _ in range(8)]
for s in streams:
    with torch.cuda.stream(s):
        x = torch.ones(1024, device=dev) * 2
torch.cuda.synchronize(dev)
bad = 0
for k in range(3):
    d = run(f"run {k + 3} (after torch side streams)")
    bad += int(not np.array_equal(d, ref))
if os.environ.get("PROBE_GLOO"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    tt = torch.tensor([1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    _ = float(tt.item())
    for k in range(3):
        d = run(f"run {k + 6} (after a gloo collective on a device tensor)")
        bad += int(not np.array_equal(d, ref))
    dist.destroy_process_group()
print("DIVERGED RUNS:", bad)
