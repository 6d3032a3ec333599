"""Two ranks (one process each, gloo, sharing the box's single GPU) against one rank: the point-sharded LM loop must
follow the single-rank trajectory.  On a multi-GPU node the collective is the library's own RCCL leg and every rank owns
a device; what is exercised here is everything else - sharding, the job-wide block pattern, the host-staged collective
leg (soslam_ba_set_host_allreduce), the acceptance test after the summed step scalars, the speculative linearisation
behind it - in real concurrent processes; the RCCL legs (the library's own and the torch.distributed callback) run with
the one rank a one-GPU box allows."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ITERS = 10     # BASELINE.json configs[3]: the 1 M-observation problem of configs[2] sharded by point
CONFIG = 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, backend="gloo", config=CONFIG, iters=ITERS, busy_torch_stream=False, leg="callback"):
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
    full = synth.generate_ba(config)
    with ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2)) as h:   # the library's own stream
        distributed.load_shard(h, full, rank, world)
        if busy_torch_stream:
            # The two ordering causes fixed in round 1, each made certain instead of likely:
            # (1) torch fills the reduce tensor asynchronously on ITS current stream - with tens of milliseconds (at least) of work queued on
            #     that stream the fill is certain to land after the library's first linearisation unless attach() waits;
            # (2) the collective has to be ordered on the stream the LIBRARY passes, not on torch's current one - torch's
            #     stream stays busy for the first iterations, so a collective enqueued there would reduce stale data.
            torch.cuda._sleep(40_000_000)
        if leg == "rccl":
            distributed.attach_rccl(h, rank, world, dev)    # the library's own ncclAllReduce on its own stream
            cb = None
        else:
            cb = distributed.attach(h, rank, world, dev)
        if busy_torch_stream:
            torch.cuda._sleep(40_000_000)
        h.iterate(iters)
        log = h.iteration_log()
        cams, pts = h.get_state()
        lib_stream_was_current = cb.stream_checks if cb is not None else []
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cost=np.array([it.cost for it in log]),
             accepted=np.array([it.accepted for it in log]), cams=cams, pts=pts,
             stream_checks=np.array(lib_stream_was_current),
             detail=np.array([[it.cost, it.candidate_cost, it.model_cost_change, it.relative_decrease, it.radius, it.step_norm,
                               it.accepted, it.valid, it.linear_iterations] for it in log]))
    dist.barrier()
    dist.destroy_process_group()


def _single_rank(config, iters):
    from stereo_orb_slam_amd import ba, synth
    full = synth.generate_ba(config)
    with ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2)) as h:
        h.load(full)
        h.iterate(iters)
        cost = np.array([it.cost for it in h.iteration_log()])
        cams, pts = h.get_state()
    return full, cost, cams, pts


def test_two_ranks_follow_the_single_rank_trajectory(tmp_path, oracle_lib):
    """BASELINE.json configs[3]: the 1 M-observation problem sharded by point over two ranks (gloo; both share the box's
    one GPU), 10 LM iterations (/root/reference/src/bundle_adjuster.cpp:116 is the call being replaced), ONE run: every
    iterate against the single-rank trajectory and the end state against the oracle at the north_star's bars."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    full, ref_cost, ref_cams, ref_pts = _single_rank(CONFIG, ITERS)
    assert full.n_obs == 1_000_000
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    if not np.array_equal(r0["detail"], r1["detail"]):
        np.set_printoptions(precision=12, linewidth=220)
        print("rank 0 log [cost cand mcc rho radius step acc valid lin]:\n", r0["detail"], "\nrank 1 log:\n", r1["detail"])
    assert np.array_equal(r0["cost"], r1["cost"]) and np.array_equal(r0["cams"], r1["cams"])   # replicas agree bitwise
    assert r0["accepted"].all()
    np.testing.assert_allclose(r0["cost"], ref_cost, rtol=1e-9)
    assert np.abs(r0["cams"] - ref_cams).max() < 1e-8
    # each rank holds its own point shard: together they are the single-rank points
    pts = np.concatenate([r0["pts"], r1["pts"]])
    assert pts.shape == ref_pts.shape and np.abs(pts - ref_pts).max() < 1e-7
    # the oracle on the unsharded problem, same fixed iteration count
    o = oracle_lib.default_options(max_iterations=ITERS, check_termination=0, num_threads=min(16, os.cpu_count() or 1))
    ocams, opts_, osum, olog = oracle_lib.solve(full.obs_cam, full.obs_pt, full.obs_uv, full.poses_cw(), full.points_f64(),
                                                full.proj_l, full.proj_r, full.cam_fixed, o)
    assert [e.accepted for e in olog] == list(r0["accepted"])
    assert r0["cost"][-1] == pytest.approx(osum.final_cost, rel=1e-5)            # north_star: 1e-5 on the final residual
    assert np.abs(r0["cams"] - ocams).max() < 1e-4                               #             1e-4 on pose parameters
    np.testing.assert_allclose(r0["cost"], [e.cost for e in olog], rtol=1e-7)


def test_ordering_against_a_busy_torch_stream(tmp_path):
    """The two stream-ordering causes met in round 1 on the DEVICE-callback leg (a torch tensor aliasing the reduce buffer,
    torch.distributed/nccl reducing it in place), each tested once with the race made certain (see _worker): torch's
    asynchronous fill of the reduce tensor, and a collective ordered on the caller's current stream instead of the
    library's.  One rank (all a one-GPU box gives RCCL), torch's current stream kept busy across attach() and the first
    iterations: a late fill would zero the first linearisation's cost, so the trajectory must still be the plain one,
    bitwise, and every collective must have run with the library's stream as torch's current stream."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    _, ref_cost, ref_cams, _ = _single_rank(2, 6)
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "nccl", 2, 6, True), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    assert np.array_equal(r0["cost"], ref_cost) and np.array_equal(r0["cams"], ref_cams)
    assert r0["stream_checks"].size >= 12 and r0["stream_checks"].all()     # two all-reduces (at least) per iteration


def test_rccl_device_allreduce_one_rank(tmp_path):
    """The RCCL leg of the callback (backend nccl, in-place all-reduce of a device sub-range on the library's own
    stream) with the one rank a one-GPU box allows: a process group of size 1 still goes through RCCL's launch path,
    its stream ordering against the library's kernels and the aliasing of the reduce buffer.  The trajectory must be the
    plain single-handle one, bitwise."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    _, ref_cost, ref_cams, _ = _single_rank(2, 6)
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "nccl", 2, 6), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    assert r0["accepted"].all()
    assert np.array_equal(r0["cost"], ref_cost) and np.array_equal(r0["cams"], ref_cams)


def test_native_rccl_leg_one_rank(tmp_path):
    """The library's OWN collective leg (soslam_ba_init_rccl: librccl bound with dlopen, ncclCommInitRank from a unique id,
    ncclAllReduce in place on the handle's stream) with the one rank a one-GPU box allows.  A collective being attached puts
    the iteration on the multi-rank path (payload all-reduce, scalar all-reduces, acceptance test after them, speculative
    linearisation behind that test): the trajectory must be the plain single-handle one, bitwise.  No Python runs inside
    an iteration on this leg."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    _, ref_cost, ref_cams, _ = _single_rank(2, 6)
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "nccl", 2, 6, False, "rccl"), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    assert r0["accepted"].all()
    assert np.array_equal(r0["cost"], ref_cost) and np.array_equal(r0["cams"], ref_cams)


def _failing_worker(rank, world, port, out_dir):
    """Rank 1's shard is malformed (a point observed twice by one camera): its set_problem fails.  Both ranks must learn of it
    through soslam_ba_agree_status and leave together instead of rank 0 waiting in the solve's first all-reduce."""
    import torch
    import torch.distributed as dist

    from stereo_orb_slam_amd import _lib, ba, distributed, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = synth.generate_ba(1)
    with ba.BundleAdjustment(ba.default_options(device=0, linear_solver=2)) as h:
        distributed.attach(h, rank, world, torch.device("cuda", 0))   # the host leg: no problem is needed for it
        shard = full.shard(rank, world)
        status = 0
        try:
            h.set_covisibility(full.covisibility_pairs())
            if rank == 1:
                k = int(np.flatnonzero(shard.obs_cam != 0)[0])     # an observation of a free camera
                dup = lambda a: np.ascontiguousarray(np.concatenate([a, a[k:k + 1]]))
                h.set_projection(shard.proj_l, shard.proj_r)
                h.set_problem(shard.n_cam, shard.n_pt, dup(shard.obs_cam), dup(shard.obs_pt), dup(shard.obs_uv), shard.cam_fixed)
            else:
                h.load(shard)
        except _lib.SoslamError as e:
            status = e.status
        agreed = h.agree_status(status)
        solved = False
        if agreed == 0:
            h.iterate(2)
            solved = True
    np.savez(os.path.join(out_dir, f"fail{rank}.npz"), status=status, agreed=agreed, solved=solved)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_rank_does_not_hang_the_others(tmp_path):
    """a set-up failure on one rank of a sharded job (here: a malformed shard) used to leave the other ranks in
    the first all-reduce of their solve for ever.  soslam_ba_agree_status (one MAX all-reduce of a status word, which
    BundleAdjuster::Optimize now calls before the solve and before the global read-back) lets every rank leave."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no fallback")
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "fail0.npz"), np.load(tmp_path / "fail1.npz")
    assert int(r0["status"]) == 0 and int(r1["status"]) != 0
    assert int(r0["agreed"]) == int(r1["agreed"]) != 0
    assert not bool(r0["solved"]) and not bool(r1["solved"])
