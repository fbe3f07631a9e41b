"""Data-parallel gradient synchronisation for the DATR step: one process per GPU, gradients
summed over RCCL (torch.distributed backend "nccl" on ROCm) and averaged.

What the reference does: wraps the model in
`torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)`
(/root/reference/main.py:156) -- the constructor broadcasts rank 0's parameters and buffers,
then 47.77 M fp32 gradients = 191 MB are all-reduced per step in 25 MB buckets, plus a graph
walk per step to find unused parameters.

What this module does instead:
  * rank 0's parameters and buffers are broadcast once at construction (coalesced per dtype), so
    ranks seeded with `seed + rank` (main.py:138) start from the same weights, as under DDP;
  * every bucket owns ONE flat fp32 buffer.  Autograd hands over its own gradient tensors
    (`.grad = None` at `zero_grad`); when a bucket is launched they are moved into the flat buffer
    with one multi-tensor copy and the buffer slices become the parameters' `.grad`, so the
    optimizer and the clip read the reduced values without a copy-out;
  * collectives are issued in a FIXED ORDER -- bucket i only after buckets 0..i-1 -- on every
    rank, whatever order the gradients arrive in locally (ranks may differ: a rank without boxes
    or pseudo labels produces fewer gradients).  A bucket whose predecessors are launched is
    launched from the autograd hook of its last-arriving gradient, so communication overlaps the
    rest of backward; whatever is left is launched from `finish()`;
  * the fixed order is the order in which rank 0's gradients became ready in the FIRST step
    (broadcast, then the buckets are rebuilt once -- what DDP's bucket rebuild does), so
    in steady state a bucket completes when its predecessors already have and nothing waits for
    `finish()`; the last-produced gradients (backbone layer2) get a small bucket of their own so
    that the reduction left over after backward is short;
  * xGMI is point-to-point (7 links x ~153 GB/s), so a ring all-reduce is per-link bound:
    fewer, larger buckets amortise launch latency better than DDP's 25 MB default -- the
    default here is 64 MB (3-4 buckets for 191 MB);
  * parameters that received no gradient on this rank in a step (the reference needs
    find_unused_parameters=True for those) contribute their zero-filled slice -- no graph walk -- and
    a flag per parameter ("some rank produced a gradient") is all-reduced with the buckets: DDP leaves
    the .grad of a GLOBALLY unused parameter None, so AdamW skips it (no weight decay, no moment decay,
    no step count).  `used_flags()` hands the flags to the own optimizer kernel as a device tensor (no
    host synchronisation; datr_amd.optim.FusedClipAdamW); `drop_unused_grads()` is the same thing for a
    stock optimizer (it reads the flags on the host).
Shared modules (the six aliased detection heads) appear once: parameters are de-duplicated by
identity, as `nn.Module.parameters()` already does.
"""
from __future__ import annotations

import time
import weakref
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


def broadcast_module_state(module: nn.Module, src: int = 0, group=None):
    """Rank `src`'s parameters and buffers to every rank, one broadcast per dtype (what the DDP
    constructor does, /root/reference/main.py:156).  Aliased tensors are sent once."""
    if not dist.is_initialized():
        return
    seen, by_dtype = set(), {}
    for t in list(module.parameters()) + list(module.buffers()):
        if id(t) in seen or t.numel() == 0:
            continue
        seen.add(id(t))
        by_dtype.setdefault((t.dtype, t.device), []).append(t.data)
    for (dtype, device), tensors in by_dtype.items():
        wire = dtype if dtype != torch.bool else torch.uint8
        flat = torch.cat([t.reshape(-1).to(wire) for t in tensors])
        dist.broadcast(flat, src=src, group=group)
        offset = 0
        for t in tensors:
            n = t.numel()
            # a dense tensor's storage order is its own (channels_last weights included):
            # reshape(-1) above walked logical order, so write back through logical order too
            t.copy_(flat[offset:offset + n].view(t.shape).to(dtype))
            offset += n


class _Bucket:
    ALIGN = 64       # elements

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
        # every slice starts on a 256-byte boundary: the multi-tensor kernels that read the gradient
        # views (fused AdamW, the clip's norm, the copy-in) take their 16-byte vector path only for
        # aligned pointers -- packed back to back, everything behind the first odd-sized tensor (a
        # [4] or [9] bias) was misaligned and AdamW ran 0.96 instead of 0.51 ms per step.  The
        # padding stays zero and travels through the all-reduce.
        offsets, offset = [], 0
        for p in params:
            offsets.append(offset)
            offset = (offset + p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = offset
        self.flat = torch.zeros(self.numel, device=device, dtype=dtype)
        self.views = []
        for p, off in zip(params, offsets):
            self.views.append(self.view(p, off))
            p.grad = None
        self.pending = len(params)
        self.work = None
        self.event = None
        self.launched = False
        # stream the bucket's last gradient was produced on (hooks of later buckets may launch
        # this one from a different stream)
        self.ready_stream = None


class GradAllReducer:
    """Usage per step:  reducer.zero_grad(); loss.backward(); reducer.finish(); clip; step."""

    def __init__(self, model: nn.Module, bucket_mb: float = 64.0, first_bucket_mb: float = 8.0,
                 process_group=None, side_params=None, broadcast: bool = True,
                 rebuild: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # RCCL averages inside the collective (no division pass over the buckets afterwards); gloo
        # has no AVG: sum, then divide in finish()
        nccl = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self._op = dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM
        if broadcast and dist.is_initialized() and self.world > 1:
            broadcast_module_state(model, 0, process_group)
        self.params = [p for p in model.parameters() if p.requires_grad]
        assert self.params, "no trainable parameters"
        self._index = {p: i for i, p in enumerate(self.params)}
        self.device, self.dtype = self.params[0].device, self.params[0].dtype
        if side_params is None:
            side_params = getattr(model, "side_stream_parameters", lambda: [])()
        self._side_ids = {id(p) for p in side_params}
        self._cap = int(bucket_mb * (1 << 20) / 4)
        self._tail_cap = int(first_bucket_mb * (1 << 20) / 4)
        # first step: gradients become ready roughly in reverse registration order
        self.buckets: List[_Bucket] = []
        self._build(list(reversed(range(len(self.params)))))
        self._arrival: Optional[List[int]] = [] if rebuild else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        # "a gradient was produced" per parameter: filled on the host while the buckets are gathered,
        # sent to the device without a synchronisation (pinned staging buffers, used in turn: the epoch
        # functions fetch the previous step's losses every step, so a buffer is free again two steps later),
        # MAX-reduced over the ranks in finish()
        n = len(self.params)
        pin = self.device.type == "cuda"
        self._used_stage = [torch.zeros(n, dtype=torch.int32, pin_memory=pin) for _ in range(3)]
        self._used_turn = 0
        self._used_events = [None] * len(self._used_stage)      # the device copy that last read each staging buffer
        self._used = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._used_work = None
        # diagnostics (bench.py --gpus N): a list here receives one record per finish() -- HIP events around the
        # wait for the outstanding all-reduces on the compute stream (their span = communication NOT hidden
        # behind backward), the host time of finish() and how many buckets were only launched there
        self.diag = None

    # -- bucket layout --------------------------------------------------------------------------
    def _build(self, order: List[int]):
        """Buckets over the parameters in `order` (expected readiness order).  Main-stream
        parameters: the last ones (<= tail capacity) form a small final bucket, the rest is cut
        into `cap`-sized buckets from the front.  Side-stream parameters share one bucket.  The
        launch order of the buckets is the position of their LAST parameter in `order`."""
        pos = {i: k for k, i in enumerate(order)}
        main = [i for i in order if id(self.params[i]) not in self._side_ids]
        side = [i for i in order if id(self.params[i]) in self._side_ids]
        groups: List[tuple] = []
        tail: List[int] = []
        size = 0
        while main and size + self.params[main[-1]].numel() <= self._tail_cap:
            i = main.pop()
            tail.append(i)
            size += self.params[i].numel()
        cur, size = [], 0
        for i in main:
            n = self.params[i].numel()
            if cur and size + n > self._cap:
                groups.append((cur, False))
                cur, size = [], 0
            cur.append(i)
            size += n
        if cur:
            groups.append((cur, False))
        if tail:
            groups.append((list(reversed(tail)), False))
        if side:
            groups.append((side, True))
        groups.sort(key=lambda g: max(pos[i] for i in g[0]))
        self.buckets = [_Bucket([self.params[i] for i in idx], self.device, self.dtype, side=s)
                        for idx, s in groups]
        self._bucket_of = {p: b for b in self.buckets for p in b.params}
        self._next = 0

    def _rebuild_from_first_step(self):
        """After the first backward: adopt rank 0's gradient arrival order (parameters that got
        no gradient go last, in reverse registration order) and re-cut the buckets.  Every rank
        calls this at the same point (first `finish()`), so the broadcast is matched."""
        seen = set(self._arrival)
        order = self._arrival + [i for i in reversed(range(len(self.params))) if i not in seen]
        self._arrival = None
        if dist.is_initialized() and self.world > 1:
            t = torch.tensor(order, dtype=torch.int64, device=self.device)
            dist.broadcast(t, src=0, group=self.group)
            order = t.tolist()
        kept = [None if p.grad is None else p.grad.clone() for p in self.params]
        self._build(order)              # new flat buffers; resets every .grad
        for p, g in zip(self.params, kept):
            p.grad = g

    # -- hooks ----------------------------------------------------------------------------------
    def _on_grad(self, p: nn.Parameter):
        if self._arrival is not None:
            self._arrival.append(self._index[p])
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending == 0:
            if b.flat.is_cuda:
                b.ready_stream = torch.cuda.current_stream(b.flat.device)
            self._launch_ready()

    def _launch_ready(self):
        """Launch, in index order, every bucket whose gradients are all there -- and stop at the
        first one that still waits: the sequence of collectives is the same on every rank."""
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _gather(self, b: _Bucket, stream=None, mark: bool = True):
        """Move the gradients autograd produced into the flat buffer with ONE multi-tensor copy and
        make the buffer slices the parameters' .grad.  (Pre-attaching the slices as .grad instead
        makes autograd ACCUMULATE into them: one add kernel per parameter, 300 launches and 1.4 ms
        per step.)  Parameters without a gradient this step keep their zeroed slice.
        `stream`: the stream the copy runs on when gradients may come from another one.
        `mark`: record "a gradient was produced" in this step's staging buffer (False when finish() re-homes
        already reduced gradients: the buffer is then on its way to the device and must not be rewritten)."""
        dst, src = [], []
        stage = self._used_stage[self._used_turn]
        for p, v in zip(b.params, b.views):
            g = p.grad
            if g is not None and mark:
                stage[self._index[p]] = 1
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
        """Gather the bucket and start its all-reduce from the CURRENT stream.  Usually that is
        the stream autograd produced the bucket's last gradient on: then there is no cross-stream
        wait at all (every such wait costs an event record on the compute stream, and a dozen of
        them per step measured ~1.9 ms).  Waits are inserted only when gradients may have been
        produced elsewhere: a bucket of side-stream parameters (the detector's image-level
        discriminator runs on its own stream, detector.py) waits for EXTRA_STREAMS / the default
        stream, and a bucket that became ready on another stream than the one it is launched
        from (ordered launching can defer it to a later hook) waits for that stream."""
        b.launched = True
        collective = self.world > 1 or FORCE_COLLECTIVES
        foreign = None
        if b.flat.is_cuda:
            dev = b.flat.device
            cur = torch.cuda.current_stream(dev)
            wait = []
            if b.side or b.pending > 0:
                wait = [torch.cuda.default_stream(dev)] + list(EXTRA_STREAMS)
            elif b.ready_stream is not None and b.ready_stream != cur:
                wait = [b.ready_stream]
            for s in wait:
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
            b.ready_stream = None
            for p in b.params:          # autograd then hands over its gradient tensor as it is
                p.grad = None
        self._next = 0
        self._used_turn = (self._used_turn + 1) % len(self._used_stage)
        ev = self._used_events[self._used_turn]
        if ev is not None:              # the copy that read this buffer two steps ago: long done, unless a caller's
            ev.synchronize()            # loop never waits for the device
            self._used_events[self._used_turn] = None
        self._used_stage[self._used_turn].zero_()

    def finish(self):
        """Call after backward: launches, in order, the buckets that are still waiting (their
        missing gradients will never come: unused parameters this step), waits for every gather /
        all-reduce and averages.  Afterwards every bucketed parameter's .grad is its slice of
        the flat buffer."""
        t_host = time.perf_counter() if self.diag is not None else 0.0
        late = len(self.buckets) - self._next
        for b in self.buckets[self._next:]:
            self._launch(b)
        self._next = len(self.buckets)
        ev0 = None
        if self.diag is not None and self._used.is_cuda:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        # every bucket is gathered: this rank's used flags are complete
        self._used.copy_(self._used_stage[self._used_turn], non_blocking=True)
        if self._used.is_cuda:
            self._used_events[self._used_turn] = torch.cuda.current_stream(self.device).record_event()
        if self.world > 1 or FORCE_COLLECTIVES:
            self._used_work = dist.all_reduce(self._used, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
            if b.event is not None:
                torch.cuda.current_stream(b.flat.device).wait_event(b.event)
                b.event = None
            if self.world > 1 and self._op == dist.ReduceOp.SUM:
                b.flat.div_(self.world)
        if self._used_work is not None:
            self._used_work.wait()
            self._used_work = None
        if self.diag is not None:
            ev1 = None
            if ev0 is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
            self.diag.append({"events": (ev0, ev1), "host_ms": (time.perf_counter() - t_host) * 1e3,
                              "buckets_launched_in_finish": late})
        if self._arrival is not None:
            self._rebuild_from_first_step()
            for b in self.buckets:      # re-home this step's (already reduced) gradients; the used flags are on
                self._gather(b, mark=False)     # their way to the device: the staging buffer is not touched

    def used_flags(self) -> torch.Tensor:
        """int32 device tensor, one entry per parameter of `self.params`: 1 = some rank produced a
        gradient for it in the step `finish()` just closed.  Valid after finish()."""
        return self._used

    def drop_unused_grads(self):
        """.grad = None for the parameters NO rank used this step -- what DistributedDataParallel with
        find_unused_parameters=True leaves behind (/root/reference/main.py:156), so a stock optimizer
        skips them.  Reads the flags on the host (a synchronisation); the own optimizer kernel takes
        `used_flags()` instead."""
        for p, u in zip(self.params, self._used.tolist()):
            if not u:
                p.grad = None

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def total_bytes(self) -> int:
        return sum(b.numel for b in self.buckets) * 4


# model -> its reducer (False = "none, and do not create one").  A weak registry rather than an
# attribute of the model or of the args bag: the reference pickles `args` into every checkpoint
# (/root/reference/main.py:401-412) and deep-copies the model for the EMA teacher (EMA.py:33) --
# neither may drag 191 MB of flat gradient buckets, hook handles and stream objects along.
_REDUCERS = weakref.WeakKeyDictionary()


def attach_reducer(model: nn.Module, reducer) -> None:
    """Registers `reducer` (a GradAllReducer, or False to forbid the on-demand one) for `model`."""
    _REDUCERS[model] = reducer


def reducer_for(model: nn.Module, args=None) -> Optional[GradAllReducer]:
    """The reducer the epoch functions (datr_amd.engine) synchronise gradients with: the one
    registered for the model (`attach_reducer`, `training.build_training`), otherwise one per model
    created on first use whenever a process group with more than one rank is up (or
    DATR_DIST_FORCE_COLLECTIVES=1) and the model is not already wrapped in
    DistributedDataParallel.  None = single process, plain `.grad`s.  (`args.reducer`, if a caller
    put one there, still wins -- but nothing in this package stores it on the args bag.)"""
    r = getattr(args, "reducer", None) if args is not None else None
    if r is not None:
        return r
    if isinstance(model, nn.parallel.DistributedDataParallel):
        return None
    r = _REDUCERS.get(model)
    if r is False:
        return None
    if r is None and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES):
        r = GradAllReducer(model)
        _REDUCERS[model] = r
    return r


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
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
        dist.barrier()
    return rank, local_rank, world
