"""Batch containers and small tensor / process-group helpers used on the hot path.

Mirror of the pieces of the reference's `util/misc.py` the training step touches
(/root/reference/util/misc.py): `NestedTensor` (:313-385), `nested_tensor_from_tensor_list`
(:387-409), `collate_fn_da` (:291-300), `inverse_sigmoid` (:587-591, eps = 1e-3), `accuracy`
(:533-549), `reduce_dict` (:139-163), `get_world_size` / `is_dist_avail_and_initialized`
(:458-469).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import Tensor


class NestedTensor:
    """Images zero-padded to a common size plus a bool mask (True = padding)."""

    def __init__(self, tensors: Tensor, mask: Optional[Tensor], padded: Optional[bool] = None):
        self.tensors = tensors
        self.mask = mask
        # host-side knowledge about the mask: False = no pixel is padding (every image already
        # had the batch's size), True = some are, None = unknown (treated as padded)
        self.padded = padded

    def to(self, device, non_blocking: bool = False) -> "NestedTensor":
        mask = None if self.mask is None else self.mask.to(device, non_blocking=non_blocking)
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), mask, self.padded)

    def decompose(self):
        return self.tensors, self.mask

    @property
    def device(self):
        return self.tensors.device

    @property
    def shape(self):
        return {"tensors.shape": self.tensors.shape,
                "mask.shape": None if self.mask is None else self.mask.shape}

    def __repr__(self):
        return f"NestedTensor(tensors={tuple(self.tensors.shape)})"


def nested_tensor_from_tensor_list(tensor_list: Sequence[Tensor]) -> NestedTensor:
    """[C,H_i,W_i] images (or one [B,C,H,W] tensor) -> padded batch + mask.
    NB a 4-d input is returned AS IS (`.tensors` aliases the caller's batch, in its memory format): the reference
    builds a fresh zero-padded copy (util/misc.py:387-409) with the same values, so only a caller that modifies
    `.tensors` in place afterwards could tell -- none on this path does (the augmentation kernels write new tensors,
    the backbone never writes its input); clone first if you must."""
    if isinstance(tensor_list, Tensor) and tensor_list.dim() == 4:
        # one [B, C, H, W] batch: every image fills it, nothing to pad -- the batch itself with an all-False mask,
        # in the memory format it came in (the reference copies image by image into a new contiguous tensor,
        # util/misc.py:387-409: the same values; a channels_last slice of a training batch -- the teacher's input in
        # engine.train_one_epoch_with_self_training -- would leave that copy in NCHW, off the NHWC kernels)
        b, _, h, w = tensor_list.shape
        return NestedTensor(tensor_list, torch.zeros((b, h, w), dtype=torch.bool, device=tensor_list.device), False)
    if isinstance(tensor_list, Tensor):
        tensor_list = list(tensor_list.unbind(0))
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    h = max(img.shape[1] for img in tensor_list)
    w = max(img.shape[2] for img in tensor_list)
    b = len(tensor_list)
    first = tensor_list[0]
    batch = torch.zeros((b, c, h, w), dtype=first.dtype, device=first.device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=first.device)
    for i, img in enumerate(tensor_list):
        batch[i, :, :img.shape[1], :img.shape[2]].copy_(img)
        mask[i, :img.shape[1], :img.shape[2]] = False
    padded = any(img.shape[1] != h or img.shape[2] != w for img in tensor_list)
    return NestedTensor(batch, mask, padded)


def collate_fn_da(batch):
    """Source images first, then target images, in ONE padded batch of 2B images."""
    src_imgs, src_labels, tgt_imgs, tgt_labels, tgt_strong = list(zip(*batch))
    samples = nested_tensor_from_tensor_list(src_imgs + tgt_imgs)
    strong = None
    if tgt_strong[0] is not None:
        strong = nested_tensor_from_tensor_list(src_imgs + tgt_strong)
    return samples, src_labels, tgt_labels, strong


def inverse_sigmoid(x: Tensor, eps: float = 1e-3) -> Tensor:
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


@torch.no_grad()
def accuracy(output: Tensor, target: Tensor, topk=(1,)) -> List[Tensor]:
    """precision@k in percent."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    hit = pred.eq(target.view(1, -1).expand_as(pred))
    return [hit[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def reduce_dict(input_dict, average: bool = True):
    """All-reduce a dict of scalar tensors (sorted keys so every rank stacks the same order)."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world
        return {k: v for k, v in zip(names, values)}
