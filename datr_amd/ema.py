"""EMA teachers for the self-training stage.

Mirror of /root/reference/models/dino/EMA.py: `ModelEMA` (:21-54, decay ramp
d(k) = decay * (1 - exp(-k / 2000)), updated once per EPOCH by /root/reference/main.py:382) and
`CosineEMA` (:92-135).  The per-tensor python loop over the 640 state_dict entries
(`v *= d; v += (1 - d) * msd[k]`, EMA.py:46-50) becomes two multi-tensor (foreach) launches per
alias multiplicity over the floating-point entries (see `_ema_update_` for the aliased heads).
"""
from __future__ import annotations

import math
from copy import deepcopy

import numpy as np
import torch
from torch import nn


def is_parallel(model) -> bool:
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def _unwrap(model):
    return model.module if is_parallel(model) else model


def copy_attr(a, b, include=(), exclude=()):
    for k, v in b.__dict__.items():
        if (len(include) and k not in include) or k.startswith("_") or k in exclude:
            continue
        setattr(a, k, v)


class _DevicePlan:
    """Device-resident tables of csrc/ema.hip for one (EMA model, model) pair: a `datr_ema_tensor`
    per distinct float32 tensor and the work list of 4096-element pieces."""

    def __init__(self, groups, device):
        from . import _native
        import numpy as np
        piece = int(_native.lib.datr_ema_piece_elements())
        tens = np.zeros((len(groups), 4), dtype=np.int64)
        pieces = []
        for i, (v, m, k) in enumerate(groups):
            tens[i] = (v.data_ptr(), m.data_ptr(), v.numel(), k)
            pieces += [(i, off) for off in range(0, v.numel(), piece)]
        self.key = tuple(map(tuple, tens.tolist()))
        self.npieces = len(pieces)
        self.tensors = torch.from_numpy(tens).to(device)
        self.pieces = torch.tensor(pieces, dtype=torch.int64).reshape(-1, 2).to(device)
        self.device = device

    def run(self, d: float):
        from . import _native
        with _native.on_device(self.device):
            rc = _native.lib.datr_ema_update_f32(self.tensors.data_ptr(), self.pieces.data_ptr(), self.npieces,
                                                 float(d), _native.current_stream_ptr(self.device))
        _native.check(rc, "ema_update")


_PLANS = {}      # id(ema_model) -> _DevicePlan


@torch.no_grad()
def _ema_update_(ema_model: nn.Module, model: nn.Module, d: float) -> None:
    """The reference walks the state_dict KEYS (EMA.py:46-50), so a tensor that appears under k
    keys -- the detection heads shared by the six decoder layers: `bbox_embed.0..5.*`,
    `class_embed.0..5.*` and their `transformer.decoder.*` aliases, 12 keys per tensor -- is
    updated k times per call.  Reproduced, not "fixed" to a single update: every distinct tensor gets
    `v *= d; v += (1. - d) * m` replayed k times, the reference's arithmetic bit for bit.  On the
    device that is ONE launch of csrc/ema.hip over all float32 tensors (tables cached per EMA model
    and rebuilt when a storage moved); tensors on the host (CPU runs) take the same arithmetic
    through torch._foreach."""
    msd = _unwrap(model).state_dict()
    first, count = {}, {}
    for k, v in ema_model.state_dict().items():
        if v.dtype.is_floating_point:
            ptr = v.data_ptr()
            if ptr not in first:
                first[ptr] = (v, msd[k].detach())
            count[ptr] = count.get(ptr, 0) + 1
    dev, host = [], {}
    for ptr, (v, m) in first.items():
        if (v.is_cuda and m.is_cuda and v.dtype == m.dtype == torch.float32 and v.is_contiguous()
                and m.is_contiguous() and v.numel() == m.numel() and v.device == m.device):
            dev.append((v, m, count[ptr]))
        else:
            dst, src = host.setdefault(count[ptr], ([], []))
            dst.append(v)
            src.append(m)
    if dev:
        key = tuple((v.data_ptr(), m.data_ptr(), v.numel(), k) for v, m, k in dev)
        plan = _PLANS.get(id(ema_model))
        if plan is None or plan.key != key:
            plan = _PLANS[id(ema_model)] = _DevicePlan(dev, dev[0][0].device)
        plan.run(d)
        # The kernel writes through raw pointers: autograd's version counters must move as they
        # would under `v *= d`, or caches keyed on (data_ptr, _version) -- the folded frozen-BN
        # weights of pointwise.fold_frozen_bn, FrozenBatchNorm2d.scale_shift -- keep serving the
        # teacher's weights from before the update.
        torch._C._increment_version([v for v, _, _ in dev])
    for k, (dst, src) in host.items():
        for _ in range(k):
            torch._foreach_mul_(dst, d)
            torch._foreach_add_(dst, torch._foreach_mul(src, 1.0 - d))


class ModelEMA:
    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(_unwrap(model)).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)

    def update(self, model):
        self.updates += 1
        _ema_update_(self.ema, model, self.decay(self.updates))

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        copy_attr(self.ema, model, include, exclude)


class CosineEMA:
    def __init__(self, model, decay_start=0.99, decay_end=0.9999, total_epoch=0):
        self.ema = deepcopy(_unwrap(model)).eval()
        self.total_epoch = total_epoch
        self.decay_start, self.decay_end = decay_start, decay_end
        self.decay = decay_start
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.updates = 0

    def update(self, model):
        _ema_update_(self.ema, model, self.decay)

    def update_decay(self, cur_epoch):
        self.decay = self.decay_end - (self.decay_end - self.decay_start) * \
            (np.cos(np.pi * cur_epoch / self.total_epoch) + 1) / 2

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        copy_attr(self.ema, model, include, exclude)
