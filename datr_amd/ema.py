"""EMA teachers for the self-training stage.

Mirror of /root/reference/models/dino/EMA.py: `ModelEMA` (:21-54, decay ramp
d(k) = decay * (1 - exp(-k / 2000)), updated once per EPOCH by /root/reference/main.py:382) and
`CosineEMA` (:92-135).  The per-tensor python loop over the 640 state_dict entries
(`v *= d; v += (1 - d) * msd[k]`, EMA.py:46-50) becomes two multi-tensor (foreach) launches over
the floating-point entries -- same arithmetic per element.
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
    msd = _unwrap(model).state_dict()
    dst, src, seen = [], [], set()
    for k, v in ema_model.state_dict().items():
        if v.dtype.is_floating_point and v.data_ptr() not in seen:    # aliased heads: once
            seen.add(v.data_ptr())
            dst.append(v)
            src.append(msd[k].detach())
    torch._foreach_mul_(dst, d)
    torch._foreach_add_(dst, src, alpha=1.0 - d)


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
