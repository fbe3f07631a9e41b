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


@torch.no_grad()
def _ema_update_(ema_model: nn.Module, model: nn.Module, d: float) -> None:
    """The reference walks the state_dict KEYS (EMA.py:46-50), so a tensor that appears under k
    keys -- the detection heads shared by the six decoder layers: `bbox_embed.0..5.*`,
    `class_embed.0..5.*` and their `transformer.decoder.*` aliases, 12 keys per tensor -- is
    updated k times per call: v <- d^k v + (1 - d^k) m.  Reproduced here as ONE update with the
    effective decay d^k per alias group (equal up to rounding), not "fixed" to a single update:
    the teacher's heads are meant to track the student the way the reference's do."""
    msd = _unwrap(model).state_dict()
    first, count = {}, {}
    for k, v in ema_model.state_dict().items():
        if v.dtype.is_floating_point:
            ptr = v.data_ptr()
            if ptr not in first:
                first[ptr] = (v, msd[k].detach())
            count[ptr] = count.get(ptr, 0) + 1
    by_k = {}
    for ptr, (v, m) in first.items():
        dst, src = by_k.setdefault(count[ptr], ([], []))
        dst.append(v)
        src.append(m)
    for k, (dst, src) in by_k.items():
        dk = d ** k
        torch._foreach_mul_(dst, dk)
        torch._foreach_add_(dst, src, alpha=1.0 - dk)


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
