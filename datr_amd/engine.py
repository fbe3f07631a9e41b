"""One burn-in training epoch: the counterpart of the reference's `engine.train_one_epoch`
(/root/reference/engine.py:29-142) with the same signature and the same per-step sequence:
H2D move, model(samples, targets), criterion, sum(loss * weight) over weight_dict keys,
reduce_dict for logging, non-finite guard, zero_grad / backward / clip_grad_norm_ / step.
The reference's MetricLogger pretty-printer is bookkeeping (out of scope); the
returned dict carries the same averaged stats keys (`loss`, every `<key>` scaled and
`<key>_unscaled`, `class_error`, `lr`)."""
from __future__ import annotations

import math
import sys
from collections import defaultdict
from typing import Iterable

import torch

from .nested import reduce_dict


def train_one_epoch(model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable,
                    optimizer: torch.optim.Optimizer, device: torch.device, epoch: int,
                    max_norm: float = 0, wo_class_error=False, lr_scheduler=None, args=None,
                    logger=None, ema_m=None):
    amp = bool(getattr(args, "amp", False))
    scaler = torch.cuda.amp.GradScaler(enabled=amp)
    need_tgt_for_training = bool(getattr(args, "use_dn", False))
    model.train()
    criterion.train()
    sums, counts = defaultdict(float), defaultdict(int)
    last = {}
    steps = 0
    for samples, targets, _, _ in data_loader:
        samples = samples.to(device)
        targets = [{k: v.to(device) for k, v in t.items()} for t in targets]
        with torch.autocast(device_type=device.type, enabled=amp):
            outputs = model(samples, targets) if need_tgt_for_training else model(samples)
            loss_dict = criterion(outputs, targets)
            weight_dict = criterion.weight_dict
            losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)

        loss_dict_reduced = reduce_dict(loss_dict)
        scaled = {k: v * weight_dict[k] for k, v in loss_dict_reduced.items() if k in weight_dict}
        loss_value = sum(scaled.values()).item()
        if not math.isfinite(loss_value):
            print(f"Loss is {loss_value}, stopping training")
            print(loss_dict_reduced)
            sys.exit(1)

        optimizer.zero_grad()
        if amp:
            scaler.scale(losses).backward()
            if max_norm > 0:
                scaler.unscale_(optimizer)
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
            scaler.step(optimizer)
            scaler.update()
        else:
            losses.backward()
            if max_norm > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
            optimizer.step()
        if getattr(args, "onecyclelr", False):
            lr_scheduler.step()
        if getattr(args, "use_ema", False) and epoch >= getattr(args, "ema_epoch", 0):
            ema_m.update(model)

        stats = {"loss": loss_value, "lr": optimizer.param_groups[0]["lr"]}
        stats.update({k: float(v) for k, v in scaled.items()})
        stats.update({f"{k}_unscaled": float(v) for k, v in loss_dict_reduced.items()})
        if "class_error" in loss_dict_reduced:
            stats["class_error"] = float(loss_dict_reduced["class_error"])
        for k, v in stats.items():
            sums[k] += v
            counts[k] += 1
        last = stats
        steps += 1
        if getattr(args, "debug", False) and steps % 15 == 0:
            print("BREAK!" * 5)
            break
    resstat = {k: sums[k] / counts[k] for k in sums if counts[k] > 0}
    resstat["_last"] = last
    return resstat
