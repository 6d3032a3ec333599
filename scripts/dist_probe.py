"""Development aid: LM trajectory of N gloo ranks sharing one GPU against the single-rank trajectory.
usage: python -m torch.distributed.run --nproc-per-node N scripts/dist_probe.py [config] [iters]   (or plain python for N = 1)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stereo_orb_slam_amd import ba, synth

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
full = synth.generate_ba(cfg)
stream = torch.cuda.current_stream(dev)
h = ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2, stream=stream.cuda_stream))
if world > 1:
    from stereo_orb_slam_amd import distributed
    prob = distributed.load_shard(h, full, rank, world)
    distributed.attach(h, rank, world, dev)
else:
    prob = full
    h.load(prob)
import numpy as np
from stereo_orb_slam_amd import _lib as L
if os.environ.get("PROBE_DUMP"):
    h.iterate(1)
    cams0, _ = h.get_state()
    if rank == 0:
        print("world", world, "cams after 1 iteration: sum", repr(float(cams0.sum())), "abs sum", repr(float(np.abs(cams0).sum())), flush=True)
    h.debug_step(3e4)
    S, rhs, dc = h.debug_read(L.DBG_S_DENSE), h.debug_read(L.DBG_RHS), h.debug_read(L.DBG_STEP_CAM)
    cams, pts = h.get_state()
    if rank == 0:
        np.savez(os.path.join(os.environ["PROBE_DUMP"], f"probe_w{world}.npz"), S=S, rhs=rhs, dc=dc, cams=cams, cams0=cams0)
    iters = 0
s = h.iterate(iters)
if rank == 0:
    for i, it in enumerate(h.iteration_log()):
        print(f"world {world} it {i}: cost {it.cost:.10e} cand {it.candidate_cost:.10e} mcc {it.model_cost_change:.10e} "
              f"rho {it.relative_decrease:.6f} radius {it.radius:.3e} step {it.step_norm:.6e} acc {it.accepted} valid {it.valid} lin {it.linear_iterations}", flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
