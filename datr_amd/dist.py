"""Data-parallel gradient synchronisation for the DATR step: one process per GPU, gradients
summed over RCCL (torch.distributed backend "nccl" on ROCm) and averaged.

What the reference does: wraps the model in
`torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)`
(/root/reference/main.py:156) -- 47.77 M fp32 gradients = 191 MB all-reduced per step in
25 MB buckets, plus a graph walk per step to find unused parameters.

What this module does instead:
  * every bucket owns ONE flat fp32 buffer and each parameter's `.grad` is a view into it, so
    there is no copy into or out of communication buffers and `zero_grad` is one memset per
    bucket;
  * buckets are filled in reverse parameter order (heads / decoder first, backbone layer2
    last), and a bucket's all-reduce is launched asynchronously from the autograd hook of its
    last-arriving gradient, so communication overlaps the rest of backward;
  * xGMI is point-to-point (7 links x ~153 GB/s), so a ring all-reduce is per-link bound:
    fewer, larger buckets amortise launch latency better than DDP's 25 MB default -- the
    default here is 64 MB (3 buckets for 191 MB), with a small FIRST bucket so that the
    reduction of the last gradients produced (backbone layer2) is short;
  * parameters that received no gradient in a step (the reference needs
    find_unused_parameters=True for those) simply keep their zero-filled view, and their
    bucket is launched from `finish()` -- no graph walk, no hang.
Shared modules (the six aliased detection heads) appear once: parameters are de-duplicated by
identity, as `nn.Module.parameters()` already does.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


# streams other than the default one on which some gradient may be produced (registered by the
# code that creates them); see GradAllReducer._launch
EXTRA_STREAMS = []
# DATR_DIST_FORCE_COLLECTIVES=1: initialise the process group and issue every collective even at
# world size 1 -- lets a 1-GPU box exercise the real RCCL code path (async all-reduce from autograd
# hooks, stream ordering); values are unchanged (sum over one rank, divided by 1)
FORCE_COLLECTIVES = __import__("os").environ.get("DATR_DIST_FORCE_COLLECTIVES", "0") == "1"


_LAUNCH_STREAMS = {}


def _launch_stream(device):
    key = (device.type, device.index)
    if key not in _LAUNCH_STREAMS:
        _LAUNCH_STREAMS[key] = torch.cuda.Stream(device=device)
    return _LAUNCH_STREAMS[key]


class _Bucket:
    def view(self, p: nn.Parameter, offset: int) -> torch.Tensor:
        """The slice of the flat buffer that is `p`'s gradient, with `p`'s own strides: a
        channels_last conv weight gets a channels_last gradient view (fused AdamW requires
        parameter and gradient layouts to match; autograd then accumulates without a
        re-layout)."""
        dense = p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last) \
            or (p.dim() == 5 and p.is_contiguous(memory_format=torch.channels_last_3d))
        if dense:
            return self.flat.as_strided(p.size(), p.stride(), offset)
        return self.flat[offset:offset + p.numel()].view_as(p)

    def __init__(self, params: List[nn.Parameter], device, dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, device=device, dtype=dtype)
        offset = 0
        for p in params:
            p.grad = self.view(p, offset)
            offset += p.numel()
        self.pending = len(params)
        self.work = None
        self.launched = False


class GradAllReducer:
    """Usage per step:  reducer.zero_grad(); loss.backward(); reducer.finish(); clip; step."""

    def __init__(self, model: nn.Module, bucket_mb: float = 64.0, first_bucket_mb: float = 8.0,
                 process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in model.parameters() if p.requires_grad]
        assert params, "no trainable parameters"
        device, dtype = params[0].device, params[0].dtype
        # gradients become ready roughly in reverse registration order
        order = list(reversed(params))
        self.buckets: List[_Bucket] = []
        cap = int(bucket_mb * (1 << 20) / 4)
        first_cap = int(first_bucket_mb * (1 << 20) / 4)
        # the LAST gradients to arrive (front of `params`) get their own small bucket
        tail: List[nn.Parameter] = []
        size = 0
        while order and size + order[-1].numel() <= first_cap:
            p = order.pop()
            tail.append(p)
            size += p.numel()
        cur: List[nn.Parameter] = []
        size = 0
        for p in order:
            if cur and size + p.numel() > cap:
                self.buckets.append(_Bucket(cur, device, dtype))
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, device, dtype))
        if tail:
            self.buckets.append(_Bucket(list(reversed(tail)), device, dtype))
        self._bucket_of = {}
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                self._bucket_of[p] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- hooks ----------------------------------------------------------------------------------
    def _on_grad(self, p: nn.Parameter):
        b = self._bucket_of[p]
        # autograd may have replaced the view (e.g. first accumulation into a None grad)
        b.pending -= 1
        if b.pending == 0 and not b.launched:
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.launched = True
        if self.world > 1 or FORCE_COLLECTIVES:
            if b.flat.is_cuda:
                # Gradients of one bucket may have been written on different streams (the model runs
                # its image-level discriminator on a side stream, detector.py).  The collective is
                # ordered after the stream it is issued from, so it is issued from a small LAUNCH
                # stream that first waits for every gradient-producing stream -- the compute streams
                # themselves are never made to wait for one another here.
                dev = b.flat.device
                launch = _launch_stream(dev)
                cur = torch.cuda.current_stream(dev)
                launch.wait_stream(cur)
                for s in [torch.cuda.default_stream(dev)] + list(EXTRA_STREAMS):
                    if s != cur and s.device == dev:
                        launch.wait_stream(s)
                with torch.cuda.stream(launch):
                    b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True)
            else:
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # -- step API -------------------------------------------------------------------------------
    def zero_grad(self):
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            b.flat.zero_()
            b.pending = len(b.params)
            b.launched = False
            offset = 0
            for p in b.params:          # re-attach views if something replaced .grad
                if p.grad is None or p.grad.data_ptr() != b.flat.data_ptr() + b.flat.element_size() * offset \
                        or p.grad.stride() != p.stride():
                    p.grad = b.view(p, offset)
                offset += p.numel()

    def finish(self):
        """Call after backward: launches buckets that still wait for gradients that will never
        come (unused parameters this step), waits for every all-reduce and averages."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            if self.world > 1:
                b.flat.div_(self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def total_bytes(self) -> int:
        return sum(b.numel for b in self.buckets) * 4


def init_distributed(backend: Optional[str] = None):
    """env:// rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), as the
    reference's `init_distributed_mode` (/root/reference/util/misc.py:487-530).  Returns
    (rank, local_rank, world_size)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():
        if backend is None:
            # "nccl" == RCCL on ROCm.  DATR_DIST_BACKEND=gloo lets several ranks share one GPU
            # (functional testing of the multi-process path on a 1-GPU box).
            backend = os.environ.get("DATR_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        elif torch.cuda.is_available():
            local_rank = local_rank % torch.cuda.device_count()
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
        dist.barrier()
    return rank, local_rank, world
