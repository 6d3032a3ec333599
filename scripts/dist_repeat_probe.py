"""Development aid: three consecutive runs from the same start (set_state between them) on N gloo ranks sharing one
GPU; every rank prints its own iteration log so that rank divergence shows.
usage: python -m torch.distributed.run --nproc-per-node 2 scripts/dist_repeat_probe.py [config] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from stereo_orb_slam_amd import ba, distributed, synth

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
full = synth.generate_ba(cfg)
stream = torch.cuda.current_stream(dev)
opts = ba.default_options(device=0, linear_solver=2, stream=stream.cuda_stream, pcg_tolerance=1e-8, verbose=1 if os.environ.get('PROBE_VERBOSE') else 0)
h = ba.BundleAdjustment(opts)
prob = distributed.load_shard(h, full, rank, world)
distributed.attach(h, rank, world, dev)   # gloo: the host-staged leg (the library stages the payload itself)
if os.environ.get("PROBE_LATE_WRITES"):
    # diagnosis: does the staged host range still change after the collective returned?
    import time
    import numpy as np

    def late(host_array, op):
        t = torch.from_numpy(host_array)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        snap = host_array.copy()
        time.sleep(0.002)
        diff = int((snap != host_array).sum())
        if diff:
            print(f"PROBE rank {rank}: {diff} of {host_array.size} staged values changed AFTER all_reduce returned", flush=True)
        return 0

    h.set_host_allreduce(late, rank, world)
poses0, points0 = prob.poses_cw(), prob.points_f64()
if os.environ.get("PROBE_WARMUP"):
    h.iterate(int(os.environ["PROBE_WARMUP"]))
for run in range(3):
    if run == 1:
        opts.pcg_tolerance = 1e-10
        h.set_options(opts)
    h.set_state(poses0, points0)
    torch.cuda.synchronize(dev)
    dist.barrier()
    h.iterate(iters)
    if os.environ.get("PROBE_CUDA_COLLECTIVE"):
        tt = torch.tensor([float(run)], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        _ = float(tt.item())
    if os.environ.get("PROBE_TORCH_OP"):      # a plain torch op on torch's stream, no collective
        tt = torch.tensor([float(run)], dtype=torch.float64, device=dev)
        _ = float((tt * 2).item())
    for i, it in enumerate(h.iteration_log()):
        print(f"run {run} rank {rank} it {i}: cost {it.cost:.10e} cand {it.candidate_cost:.10e} acc {it.accepted} valid {it.valid} "
              f"lin {it.linear_iterations}", flush=True)
dist.barrier()
dist.destroy_process_group()
