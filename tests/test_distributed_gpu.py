"""Two ranks (one process each, gloo, sharing the box's single GPU) against one rank: the point-sharded LM loop must
follow the single-rank trajectory.  On a multi-GPU node the backend is nccl (= RCCL) and every rank owns a device;
what is exercised here is everything else - sharding, the job-wide block pattern, the all-reduce callback on a torch
tensor aliasing the library's reduce buffer, its ordering against the library's own stream, the summed step
scalars - in real concurrent processes."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ITERS = 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, backend="gloo"):
    import torch
    import torch.distributed as dist

    from stereo_orb_slam_amd import ba, distributed, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    full = synth.generate_ba(2)
    with ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2)) as h:   # the library's own stream
        distributed.load_shard(h, full, rank, world)
        distributed.attach(h, rank, world, dev)
        h.iterate(ITERS)
        log = h.iteration_log()
        cams, _ = h.get_state()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cost=np.array([it.cost for it in log]),
             accepted=np.array([it.accepted for it in log]), cams=cams,
             detail=np.array([[it.cost, it.candidate_cost, it.model_cost_change, it.relative_decrease, it.radius, it.step_norm,
                               it.accepted, it.valid, it.linear_iterations] for it in log]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_follow_the_single_rank_trajectory(tmp_path):
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    from stereo_orb_slam_amd import ba, synth

    full = synth.generate_ba(2)
    with ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2)) as h:
        h.load(full)
        h.iterate(ITERS)
        ref_cost = np.array([it.cost for it in h.iteration_log()])
        ref_cams, _ = h.get_state()
    for attempt in range(3):   # an ordering bug shows up intermittently: several independent launches
        out = tmp_path / f"run{attempt}"
        out.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
        r0, r1 = np.load(out / "rank0.npz"), np.load(out / "rank1.npz")
        if not np.array_equal(r0["detail"], r1["detail"]):
            np.set_printoptions(precision=12, linewidth=220)
            print("rank 0 log [cost cand mcc rho radius step acc valid lin]:\n", r0["detail"], "\nrank 1 log:\n", r1["detail"])
        assert np.array_equal(r0["cost"], r1["cost"]) and np.array_equal(r0["cams"], r1["cams"])   # replicas agree bitwise
        assert r0["accepted"].all()
        np.testing.assert_allclose(r0["cost"], ref_cost, rtol=1e-9)
        assert np.abs(r0["cams"] - ref_cams).max() < 1e-8


def test_rccl_device_allreduce_one_rank(tmp_path):
    """The RCCL leg of the callback (backend nccl, in-place all-reduce of a device sub-range on the library's own
    stream) with the one rank a one-GPU box allows: a process group of size 1 still goes through RCCL's launch path,
    its stream ordering against the library's kernels and the aliasing of the reduce buffer.  The trajectory must be the
    plain single-handle one, bitwise."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    from stereo_orb_slam_amd import ba, synth

    full = synth.generate_ba(2)
    with ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2)) as h:
        h.load(full)
        h.iterate(ITERS)
        ref_cost = np.array([it.cost for it in h.iteration_log()])
        ref_cams, _ = h.get_state()
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "nccl"), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    assert r0["accepted"].all()
    assert np.array_equal(r0["cost"], ref_cost) and np.array_equal(r0["cams"], ref_cams)
