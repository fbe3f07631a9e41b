"""Training set-up of the hot path: what the reference's `main.py` does between "fix the seed"
and the epoch loop (/root/reference/main.py:137-165) -- seed + rank, build model / criterion /
post-processors through the registry, move to the device, gradient synchronisation over the
ranks, 'default' parameter groups, AdamW -- with this build's device-side choices made in one
place: the backbone in NHWC (`torch.channels_last`, the layout MIOpen's fastest fp32 solvers
and the own conv kernels want), the per-shape library kernel selections of `datr_amd.tuning`,
clip + AdamW as multi-tensor launches of the own kernels (`datr_amd.optim`, same update rule), and
`datr_amd.dist.GradAllReducer` in place of `DistributedDataParallel` (main.py:156).

`bench.py`, the GPU tests and the tools under `tools/` all obtain their training state here and
run it through `datr_amd.engine.train_one_epoch*`: the step that is measured is the step the
epoch functions execute."""
from __future__ import annotations

import argparse
from typing import Optional

import torch

from . import detector  # noqa: F401  (registers 'dino' in MODULE_BUILD_FUNCS)
from .config import c2f_args, get_param_dict
from .dist import FORCE_COLLECTIVES, GradAllReducer, attach_reducer
from .nested import NestedTensor
from .registry import MODULE_BUILD_FUNCS


def build_training(cfg: Optional[argparse.Namespace] = None, device="cuda", *, seed: int = 0,
                   rank: int = 0, channels_last: bool = True, tuned_gemm: bool = True,
                   fused_optimizer: bool = True, reducer: Optional[bool] = None):
    """Returns a Namespace(model, criterion, postprocessors, optimizer, reducer, cfg).
    `reducer`: True / False force the flat-bucket gradient reducer on / off; None = on whenever a
    process group with more than one rank is up (or DATR_DIST_FORCE_COLLECTIVES=1).  The reducer
    is registered for the model (`dist.attach_reducer`), where the epoch functions look it up --
    NOT stored on `cfg`, which the checkpoint format pickles (`'args': args`, main.py:401-412)."""
    import torch.distributed as dist
    device = torch.device(device)
    if cfg is None:
        cfg = c2f_args(device=str(device))
    if tuned_gemm and device.type == "cuda":
        from . import tuning
        tuning.enable()              # per-shape hipBLASLt / MIOpen kernel selection, lookup only
    torch.manual_seed(seed + rank)   # main.py:138; the reducer's constructor broadcasts rank 0's
    model, criterion, postprocessors = MODULE_BUILD_FUNCS.get(cfg.modelname)(cfg)
    model.to(device)
    if channels_last and device.type == "cuda":
        model.backbone.to(memory_format=torch.channels_last)
    model.train()
    criterion.train()
    if fused_optimizer and device.type == "cuda":
        # AdamW + the gradient clip as multi-tensor launches of the own kernels (same update rule, same
        # state_dict; csrc/adamw.hip)
        from .optim import FusedClipAdamW
        optimizer = FusedClipAdamW(get_param_dict(cfg, model), lr=cfg.lr, weight_decay=cfg.weight_decay)
    else:
        optimizer = torch.optim.AdamW(get_param_dict(cfg, model), lr=cfg.lr, weight_decay=cfg.weight_decay)
    if reducer is None:
        reducer = dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)
    red = GradAllReducer(model) if reducer else None
    attach_reducer(model, red if red is not None else False)
    return argparse.Namespace(model=model, criterion=criterion, postprocessors=postprocessors,
                              optimizer=optimizer, reducer=red, cfg=cfg, device=device)


def synthetic_batch(batch_size, height, width, num_gt, device, seed, channels_last=True,
                    pad_to=None, source_only=False):
    """images randn [2B,3,H,W] (already 'normalised'); per source image `num_gt`
    boxes, labels in 1..8, cxcy ~ U(0.2,0.8), wh ~ U(0.05,0.25).  Returned in the epoch
    functions' batch format `(samples, targets, _, _)` pieces: (NestedTensor, tuple of dicts).
    `pad_to=(H', W')`: the images sit in the top-left corner of a larger zero-padded batch
    tensor with the collate function's mask (True on padding), as a batch of different-sized
    images would -- the padded (general) path of the model instead of the no-padding fast one.
    `source_only`: B source images and no target images (BASELINE configs 1-2: the model's
    `domain_adaptation = False` step); the source images and targets are the same as in the pair."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(2 * batch_size, 3, height, width, generator=g)
    if source_only:
        imgs = imgs[:batch_size].clone()
    H, W = (height, width) if pad_to is None else pad_to
    padded = (H, W) != (height, width)
    if padded:
        full = torch.zeros(imgs.shape[0], 3, H, W)
        full[:, :, :height, :width] = imgs
        mask = torch.ones(imgs.shape[0], H, W, dtype=torch.bool)
        mask[:, :height, :width] = False
        imgs = full
    else:
        mask = torch.zeros(imgs.shape[0], H, W, dtype=torch.bool)
    targets = []
    for _ in range(batch_size):
        cxcy = torch.rand(num_gt, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(num_gt, 2, generator=g) * 0.2 + 0.05
        targets.append({"boxes": torch.cat([cxcy, wh], 1).to(device),
                        "labels": torch.randint(1, 9, (num_gt,), generator=g).to(device)})
    imgs = imgs.to(device)
    if channels_last and imgs.is_cuda:
        imgs = imgs.contiguous(memory_format=torch.channels_last)
    # equal-size images: no padded pixel, which the collate function would have recorded
    # (datr_amd.nested.nested_tensor_from_tensor_list sets `padded` from the image sizes)
    return NestedTensor(imgs, mask.to(device), padded=padded), tuple(targets)


def run_steps(state, batches, epoch: int = 0):
    """`batches`: iterable of (samples, targets) -- runs them through
    `datr_amd.engine.train_one_epoch` exactly as a data loader's batches would be."""
    from .engine import train_one_epoch
    loader = ((s, t, None, None) for s, t in batches)
    return train_one_epoch(state.model, state.criterion, loader, state.optimizer, state.device,
                           epoch, state.cfg.clip_max_norm, args=state.cfg)


class Stepper:
    """`build_training(...)` + `.step(samples, targets)` = one iteration of the epoch function;
    what the profiling tools under `tools/` drive."""

    def __init__(self, device="cuda", **kw):
        self.state = build_training(device=device, **kw)
        self.model, self.criterion = self.state.model, self.state.criterion
        self.optimizer, self.reducer = self.state.optimizer, self.state.reducer

    def step(self, samples, targets):
        return run_steps(self.state, [(samples, tuple(targets))])
