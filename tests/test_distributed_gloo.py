"""The N > 1 path on CPU: world_size 2 over gloo.  There is no GPU here, so the per-rank compute is the
oracle's shard payload; what is under test is everything around it that bench.py --gpus N relies on:
the point sharding, the job-wide block pattern, the all-reduce callback object of
stereo_orb_slam_amd/distributed.py (raw pointer -> tensor slice -> torch.distributed), and the fact that
summing the shards' payloads reproduces the unsharded reduced camera system."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from stereo_orb_slam_amd import _lib, synth
        from stereo_orb_slam_amd.distributed import TorchAllReduce, TorchHostAllReduce

        full = synth.generate_ba(None, n_cam=12, n_pt=600, track_mode=1, track_len=5, spacing=0.8)
        cams, pts = full.poses_cw(), full.points_f64()
        radius = 1e4
        # 1. sharding: ranks partition points and observations; the block pattern is job-wide
        shard = full.shard(rank, world)
        counts = torch.tensor([shard.n_pt, shard.n_obs], dtype=torch.int64)
        dist.all_reduce(counts)
        assert counts.tolist() == [full.n_pt, full.n_obs]
        pairs = full.covisibility_pairs()
        local_pairs = shard.covisibility_pairs()
        assert set(map(tuple, local_pairs.tolist())) <= set(map(tuple, pairs.tolist()))

        # 2. this rank's payload, laid out like the library's reduce buffer: [S | rhs | diag | cost]
        part = oracle.shard_system(rank, world, full.obs_cam, full.obs_pt, full.obs_uv, cams, pts, full.proj_l, full.proj_r,
                                   full.cam_fixed, radius)
        n6 = len(part["rhs"])
        local_cost = oracle.cost(shard.obs_cam, shard.obs_pt, shard.obs_uv, cams, shard.points_f64(), full.proj_l, full.proj_r)
        buf = torch.zeros(n6 * n6 + 2 * n6 + 1 + 6, dtype=torch.float64)
        buf[: n6 * n6] = torch.from_numpy(part["S"].reshape(-1))
        buf[n6 * n6: n6 * n6 + n6] = torch.from_numpy(part["rhs"])
        buf[n6 * n6 + n6: n6 * n6 + 2 * n6] = torch.from_numpy(part["diag"])
        buf[n6 * n6 + 2 * n6] = local_cost
        buf[-6:] = torch.tensor([1.0 + rank, 2.0, 3.0, 4.0, 5.0, 10.0 * (rank + 1)])   # "step scalars" + a max slot

        # 3. the callback object bench.py registers, driven exactly as the C library drives it
        cb = TorchAllReduce(buf)
        main = n6 * n6 + 2 * n6 + 1
        assert cb(buf.data_ptr(), main, _lib.REDUCE_SUM, 0) == 0
        assert cb(buf.data_ptr() + 8 * main, 5, _lib.REDUCE_SUM, 0) == 0
        assert cb(buf.data_ptr() + 8 * (main + 5), 1, _lib.REDUCE_MAX, 0) == 0
        assert cb(buf.data_ptr() + 8 * (main + 6), 1, _lib.REDUCE_SUM, 0) == 1      # out of range -> error status
        assert cb(buf.data_ptr() + 4, 1, _lib.REDUCE_SUM, 0) == 1                   # misaligned
        assert cb.calls == 3

        # 3b. the host leg (soslam_ba_set_host_allreduce): the library hands over a host range it staged itself
        hcb = TorchHostAllReduce()
        stage = np.array([1.0 + rank, -2.0, 10.0 * (rank + 1)])
        assert hcb(stage[:2], _lib.REDUCE_SUM) == 0 and hcb(stage[2:], _lib.REDUCE_MAX) == 0
        assert stage.tolist() == [1.0 + 2.0, -2.0 * world, 10.0 * world] and hcb.calls == 2

        # 4. reduced payload + camera damping from the REDUCED diagonal == the unsharded system
        S = buf[: n6 * n6].numpy().reshape(n6, n6).copy()
        rhs = buf[n6 * n6: n6 * n6 + n6].numpy()
        diag = buf[n6 * n6 + n6: n6 * n6 + 2 * n6].numpy()
        s2 = (1.0 / (1.0 + np.sqrt(diag))) ** 2
        S[np.diag_indices(n6)] += np.clip(s2 * diag, 1e-6, 1e32) / (radius * s2)
        ref = oracle.step(full.obs_cam, full.obs_pt, full.obs_uv, cams, pts, full.proj_l, full.proj_r, full.cam_fixed, radius)
        np.testing.assert_allclose(S, ref["S"], rtol=1e-10, atol=1e-10 * np.abs(ref["S"]).max())
        np.testing.assert_allclose(rhs, ref["rhs"], rtol=1e-10, atol=1e-10 * np.abs(ref["rhs"]).max())
        assert float(buf[n6 * n6 + 2 * n6]) == pytest.approx(ref["cost"], rel=1e-12)
        assert buf[-6:].tolist() == [1.0 + 2.0, 4.0, 6.0, 8.0, 10.0, 10.0 * world]
        # every rank now solves the same system: identical camera step
        dc = np.linalg.solve(S, rhs)
        np.testing.assert_allclose(dc, ref["dc"][full.cam_fixed == 0].reshape(-1), rtol=1e-6, atol=1e-10)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))
