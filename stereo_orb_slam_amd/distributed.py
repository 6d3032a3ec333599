"""Multi-GPU plumbing: one process per GPU, observations sharded by point, torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm) for the one exchange step the path has.

Per LM iteration every rank builds its shard's share of the reduced camera system; the shares are summed
with ONE all-reduce of [block-sparse S | rhs | diag(B) | g_c | cost] (2.7 MB at BASELINE.json configs[2..3]),
every rank then solves the same small system, back-substitutes its own points, and five step scalars are
summed with a second, 40-byte all-reduce.  The library stays collective-agnostic: it calls back into
``TorchAllReduce`` with a device pointer that lies inside a torch tensor this module allocated for it.
"""
from __future__ import annotations

import torch

from . import _lib


class TorchAllReduce:
    """The soslam_allreduce_fn of include/soslam_ba.h, backed by torch.distributed.

    ``tensor`` is the f64 buffer handed to ``soslam_ba_set_reduce_buffer``; the library asks for in-place
    reductions of sub-ranges of it by raw pointer.  Works for any backend/device (gloo + CPU tensors in the
    tests, nccl + GPU tensors in bench.py)."""

    def __init__(self, tensor: torch.Tensor, group=None):
        assert tensor.dtype == torch.float64 and tensor.is_contiguous()
        self.tensor = tensor
        self.base = tensor.data_ptr()
        self.count = tensor.numel()
        self.group = group
        self.calls = 0
        self.elements = 0
        self.stream_checks = []    # per call: the stream the library passed was torch's current stream during the collective

    def __call__(self, ptr: int, count: int, op: int, stream: int) -> int:
        import torch.distributed as dist
        off = (ptr - self.base) // 8
        if (ptr - self.base) % 8 or off < 0 or off + count > self.count:
            return 1
        # The process group orders its collective after the pending work of torch's CURRENT stream and that stream's
        # later work after the collective.  The library's kernels run on the stream it passes here - its own
        # non-blocking stream unless soslam_ba_options.stream named another - so that stream is made current for
        # the call; relying on the caller's current stream would be a race whenever the two differ.
        view = self.tensor[off:off + count]
        rop = dist.ReduceOp.SUM if op == _lib.REDUCE_SUM else dist.ReduceOp.MAX
        if self.tensor.is_cuda:
            dev = self.tensor.device
            lib_stream = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
            with torch.cuda.stream(lib_stream):
                self.stream_checks.append(torch.cuda.current_stream(dev).cuda_stream == (stream or 0))
                dist.all_reduce(view, op=rop, group=self.group)
        else:
            dist.all_reduce(view, op=rop, group=self.group)
        self.calls += 1
        self.elements += count
        return 0


class TorchHostAllReduce:
    """The soslam_host_allreduce_fn of include/soslam_ba.h, backed by torch.distributed on HOST memory: the library stages
    the range through its own pinned buffer on its own stream and hands over a host array; no torch device tensor, stream
    or allocator takes part (backends without a device path: gloo in the tests and rehearsals)."""

    def __init__(self, group=None):
        self.group = group
        self.calls = 0
        self.elements = 0
        self.stream_checks = []

    def __call__(self, host_array, op: int) -> int:
        import torch.distributed as dist
        t = torch.from_numpy(host_array)        # shares the library's staging memory
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == _lib.REDUCE_SUM else dist.ReduceOp.MAX, group=self.group)
        self.calls += 1
        self.elements += host_array.size
        return 0


def attach(handle, rank: int, world: int, device, group=None):
    """Give a loaded BundleAdjustment handle the torch.distributed collective as its all-reduce callback.

    Backend nccl (= RCCL): a torch-owned reduce buffer reduced in place on the device (TorchAllReduce).  Any other backend:
    the host leg (TorchHostAllReduce) - the library stages the payload itself."""
    import torch.distributed as dist
    if dist.get_backend(group) != "nccl":
        cb = TorchHostAllReduce(group)
        handle.set_host_allreduce(cb, rank, world)
        return cb
    n = handle.reduce_buffer_count()
    t = torch.zeros(n, dtype=torch.float64, device=device)
    if t.is_cuda:
        # torch fills the tensor asynchronously on ITS current stream; the library writes into it from its own stream.
        # Without this wait the fill can land after the first linearisation has put its cost there.
        torch.cuda.current_stream(t.device).synchronize()
    handle.set_reduce_buffer(t.data_ptr(), n)
    cb = TorchAllReduce(t, group)
    handle.set_allreduce(cb, rank, world)
    handle._reduce_tensor = t   # keep alive as long as the handle
    return cb


def attach_rccl(handle, rank: int, world: int, device, group=None) -> None:
    """The library's own RCCL leg instead of the callback: rank 0 draws the unique id, torch.distributed only carries its
    128 bytes to the other ranks (the side channel; any would do), then every rank joins soslam_ba_init_rccl.  From then
    on no Python runs inside an LM iteration."""
    import torch.distributed as dist

    from . import ba
    # the id travels as a tensor on whatever device the process group's backend moves (nccl: the GPU, gloo: the host)
    carrier = device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    raw = ba.rccl_unique_id() if rank == 0 else bytes(_lib.RCCL_UNIQUE_ID_BYTES)
    t = torch.tensor(list(raw), dtype=torch.uint8, device=carrier)
    if world > 1:
        dist.broadcast(t, src=0, group=group)
    handle.init_rccl(bytes(t.cpu().tolist()), rank, world)


def load_shard(handle, full_problem, rank: int, world: int):
    """Upload rank's shard of ``full_problem`` with the job-wide block pattern; returns the shard."""
    shard = full_problem.shard(rank, world)
    handle.set_covisibility(full_problem.covisibility_pairs())
    handle.load(shard)
    return shard
