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

    def __init__(self, params: List[nn.Parameter], device, dtype, side: bool = False):
        self.params = params
        self.side = side
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.zeros(self.numel, device=device, dtype=dtype)
        self.views = []
        offset = 0
        for p in params:
            self.views.append(self.view(p, offset))
            offset += p.numel()
            p.grad = None
        self.pending = len(params)
        self.work = None
        self.event = None
        self.launched = False


class GradAllReducer:
    """Usage per step:  reducer.zero_grad(); loss.backward(); reducer.finish(); clip; step."""

    def __init__(self, model: nn.Module, bucket_mb: float = 64.0, first_bucket_mb: float = 8.0,
                 process_group=None, side_params=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # RCCL averages inside the collective (no division pass over the buckets afterwards); gloo
        # has no AVG: sum, then divide in finish()
        nccl = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self._op = dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM
        params = [p for p in model.parameters() if p.requires_grad]
        assert params, "no trainable parameters"
        device, dtype = params[0].device, params[0].dtype
        if side_params is None:
            side_params = getattr(model, "side_stream_parameters", lambda: [])()
        side_ids = {id(p) for p in side_params}
        side = [p for p in params if id(p) in side_ids]
        params = [p for p in params if id(p) not in side_ids]
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
        if side:                         # parameters whose gradients are produced on a side stream
            self.buckets.append(_Bucket(side, device, dtype, side=True))
        self._bucket_of = {}
        self._hooks = []
        for b in self.buckets:
            for p in b.params:
                self._bucket_of[p] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- hooks ----------------------------------------------------------------------------------
    def _on_grad(self, p: nn.Parameter):
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending == 0 and not b.launched:
            self._launch(b)

    def _gather(self, b: _Bucket, stream=None):
        """Move the gradients autograd produced into the flat buffer with ONE multi-tensor copy and
        make the buffer slices the parameters' .grad.  (Pre-attaching the slices as .grad instead
        makes autograd ACCUMULATE into them: one add kernel per parameter, 300 launches and 1.4 ms
        per step.)  Parameters without a gradient this step keep their zeroed slice.
        `stream`: the stream the copy runs on when gradients may come from another one."""
        dst, src = [], []
        for p, v in zip(b.params, b.views):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                if g.shape != v.shape or g.dtype != v.dtype or g.device != v.device:
                    g = g.to(device=v.device, dtype=v.dtype).expand_as(v)
                dst.append(v)
                src.append(g)
                if stream is not None and g.is_cuda:
                    g.record_stream(stream)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(b.params, b.views):
            p.grad = v

    def _launch(self, b: _Bucket):
        """Gather the bucket and start its all-reduce from the CURRENT stream (the one autograd
        produced the bucket's last gradient on).  No helper stream: every cross-stream wait costs
        an event record on the compute stream, and a dozen of them per step measured ~1.9 ms.
        Only a bucket of side-stream parameters (the detector's image-level discriminator runs on
        its own stream, detector.py) may see gradients from another stream; that bucket first
        waits for the streams in EXTRA_STREAMS / the default stream."""
        b.launched = True
        collective = self.world > 1 or FORCE_COLLECTIVES
        foreign = None
        if b.flat.is_cuda:
            dev = b.flat.device
            cur = torch.cuda.current_stream(dev)
            if b.side:
                for s in [torch.cuda.default_stream(dev)] + list(EXTRA_STREAMS):
                    if s != cur and s.device == dev:
                        cur.wait_stream(s)
                foreign = cur
        self._gather(b, foreign)
        if collective:
            b.work = dist.all_reduce(b.flat, op=self._op, group=self.group, async_op=True)
        elif b.flat.is_cuda and cur != torch.cuda.default_stream(dev):
            b.event = cur.record_event()

    # -- step API -------------------------------------------------------------------------------
    def zero_grad(self):
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            b.event = None
            b.flat.zero_()
            b.pending = len(b.params)
            b.launched = False
            for p in b.params:          # autograd then hands over its gradient tensor as it is
                p.grad = None

    def finish(self):
        """Call after backward: launches buckets that still wait for gradients that will never
        come (unused parameters this step), waits for every gather / all-reduce and averages.
        Afterwards every bucketed parameter's .grad is its slice of the flat buffer."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            if b.event is not None:
                torch.cuda.current_stream(b.flat.device).wait_event(b.event)
                b.event = None
            if self.world > 1 and self._op == dist.ReduceOp.SUM:
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
