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
        self.host_staged = False   # set by attach() for backends that stage device tensors through the host (gloo)
        self.calls = 0
        self.elements = 0

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
                if self.host_staged:
                    # rehearsal backends without a device path of their own (gloo): stage through the host here,
                    # synchronously - wait for the library's kernels, reduce a host copy, copy back, wait again
                    lib_stream.synchronize()
                    host = view.cpu()
                    dist.all_reduce(host, op=rop, group=self.group)
                    view.copy_(host)
                    torch.cuda.synchronize(dev)
                else:
                    dist.all_reduce(view, op=rop, group=self.group)
        else:
            dist.all_reduce(view, op=rop, group=self.group)
        self.calls += 1
        self.elements += count
        return 0


def attach(handle, rank: int, world: int, device, group=None) -> TorchAllReduce:
    """Give a loaded BundleAdjustment handle a torch-owned reduce buffer and the all-reduce callback."""
    n = handle.reduce_buffer_count()
    t = torch.zeros(n, dtype=torch.float64, device=device)
    if t.is_cuda:
        # torch fills the tensor asynchronously on ITS current stream; the library writes into it from its own stream.
        # Without this wait the fill can land after the first linearisation has put its cost there.
        torch.cuda.current_stream(t.device).synchronize()
    handle.set_reduce_buffer(t.data_ptr(), n)
    cb = TorchAllReduce(t, group)
    import torch.distributed as dist
    cb.host_staged = t.is_cuda and dist.get_backend(group) != "nccl"
    handle.set_allreduce(cb, rank, world)
    handle._reduce_tensor = t   # keep alive as long as the handle
    return cb


def load_shard(handle, full_problem, rank: int, world: int):
    """Upload rank's shard of ``full_problem`` with the job-wide block pattern; returns the shard."""
    shard = full_problem.shard(rank, world)
    handle.set_covisibility(full_problem.covisibility_pairs())
    handle.load(shard)
    return shard
