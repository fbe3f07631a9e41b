"""Checkpoint I/O in the reference's format.

The reference saves `{'model', 'optimizer', 'lr_scheduler', 'epoch', 'args'[, 'ema_model']}` to
`checkpoint.pth` and `{'ema_model', 'epoch'}` to `best_ema_teacher.pth` / `best_ema_model.pth`
(/root/reference/main.py:396-412, :487-507), and reloads them with the DDP `module.` prefix
stripped (`clean_state_dict`, /root/reference/util/misc.py:593-599; main.py:226-271).
datr_amd's modules use the reference's parameter names, so published DATR / DINO checkpoints
load with `strict=True` (torchvision's `num_batches_tracked` entries are dropped by
FrozenBatchNorm2d, backbone.py:52-60).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Iterable, Optional

import torch


def clean_state_dict(state_dict):
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def load_model_state(model: torch.nn.Module, checkpoint, key: Optional[str] = None,
                     ignore_keywords: Iterable[str] = (), strict: bool = True):
    """`checkpoint`: a path or an already loaded dict.  `key`: 'model' / 'ema_model' (default:
    whichever is present, 'model' first).  `ignore_keywords`: the reference's
    --finetune_ignore substrings (main.py:253-262), which also turn `strict` off."""
    if isinstance(checkpoint, (str, bytes)):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    if key is None:
        key = "model" if "model" in checkpoint else ("ema_model" if "ema_model" in checkpoint else None)
    sd = checkpoint[key] if key is not None else checkpoint
    sd = clean_state_dict(sd)
    ignore = list(ignore_keywords)
    if ignore:
        sd = OrderedDict((k, v) for k, v in sd.items() if not any(w in k for w in ignore))
        strict = False
    return model.load_state_dict(sd, strict=strict)


def save_checkpoint(path, model, optimizer=None, lr_scheduler=None, epoch=0, args=None,
                    ema_model=None):
    """checkpoint.pth layout of main.py:401-412 (model saved without a DDP wrapper)."""
    if getattr(args, "reducer", None) is not None:
        # run-time objects are not configuration: a caller-supplied gradient reducer (flat buckets,
        # hook handles, streams) must not be pickled with the args bag
        import copy
        args = copy.copy(args)
        args.reducer = None
    weights = {"model": model.state_dict(), "epoch": epoch, "args": args}
    if optimizer is not None:
        weights["optimizer"] = optimizer.state_dict()
    if lr_scheduler is not None:
        weights["lr_scheduler"] = lr_scheduler.state_dict()
    if ema_model is not None:
        weights["ema_model"] = ema_model.state_dict()
    torch.save(weights, path)


def save_ema_checkpoint(path, ema_model, epoch):
    """best_ema_teacher.pth / best_ema_model.pth layout (main.py:487-507)."""
    torch.save({"ema_model": ema_model.state_dict(), "epoch": epoch}, path)


def resume(checkpoint, model, optimizer=None, lr_scheduler=None, ema_model=None, eval_only: bool = False) -> int:
    """The reference's `--resume` block (main.py:226-245): model weights from 'model' (strict), the
    EMA copy from 'ema_model' with the DDP prefix stripped when both exist, and -- unless evaluating
    -- optimizer, scheduler and epoch when all three are present.  Returns the epoch to start from
    (`checkpoint['epoch'] + 1`, or 0)."""
    if isinstance(checkpoint, (str, bytes)):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    model.load_state_dict(clean_state_dict(checkpoint["model"]))
    if ema_model is not None and "ema_model" in checkpoint:
        ema_model.load_state_dict(clean_state_dict(checkpoint["ema_model"]))
    start_epoch = 0
    if (not eval_only and optimizer is not None and lr_scheduler is not None
            and all(k in checkpoint for k in ("optimizer", "lr_scheduler", "epoch"))):
        optimizer.load_state_dict(checkpoint["optimizer"])
        lr_scheduler.load_state_dict(checkpoint["lr_scheduler"])
        start_epoch = checkpoint["epoch"] + 1
    return start_epoch
