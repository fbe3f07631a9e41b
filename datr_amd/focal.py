"""Fused sigmoid focal loss (HIP, libdatr_hip.so) behind an autograd Function.

Replaces `sigmoid_focal_loss` (/root/reference/models/dino/utils.py:79-104) plus the one-hot
construction of `SetCriterion.loss_labels` (/root/reference/models/dino/dino.py:517-526): the
target is the matched class index per row and G groups (decoder layers) go through one launch.
Device tensors only -- like the MSDA op there is no CPU implementation in the product.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native


def _focal_forward(logits, target, alpha, gamma):
    G, R, C = logits.shape
    sums = torch.empty(G, dtype=torch.float32, device=logits.device)
    scratch = torch.empty(max(int(_native.lib.datr_focal_scratch_floats(G, R)), 1),
                          dtype=torch.float32, device=logits.device)
    with _native.on_device(logits.device):
        rc = _native.lib.datr_focal_loss_forward_f32(
            logits.data_ptr(), target.data_ptr(), G, R, C, alpha, gamma, scratch.data_ptr(),
            sums.data_ptr(), _native.current_stream_ptr(logits.device))
    _native.check(rc, "focal_loss_forward")
    return sums


def _focal_backward(logits, target, grad_sums, alpha, gamma):
    G, R, C = logits.shape
    grad = torch.empty_like(logits)
    with _native.on_device(logits.device):
        rc = _native.lib.datr_focal_loss_backward_f32(
            logits.data_ptr(), target.data_ptr(), grad_sums.data_ptr(), G, R, C, alpha, gamma,
            grad.data_ptr(), _native.current_stream_ptr(logits.device))
    _native.check(rc, "focal_loss_backward")
    return grad


class _FocalSums(Function):
    @staticmethod
    def forward(ctx, logits, target, alpha, gamma):
        ctx.alpha, ctx.gamma = float(alpha), float(gamma)
        ctx.save_for_backward(logits, target)
        return _focal_forward(logits, target, ctx.alpha, ctx.gamma)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sums):
        logits, target = ctx.saved_tensors
        g = _focal_backward(logits, target, grad_sums.contiguous().float(), ctx.alpha, ctx.gamma)
        return g, None, None, None


def sigmoid_focal_loss_sums(logits: torch.Tensor, target: torch.Tensor, alpha: float = 0.25,
                            gamma: float = 2.0) -> torch.Tensor:
    """logits [G, R, C] fp32, target [G, R] int64 (class index, anything outside [0,C) = no
    positive) -> [G] sums of the element-wise focal loss."""
    if not logits.is_cuda:
        raise RuntimeError("sigmoid_focal_loss_sums: Not implemented on the CPU")
    if logits.dtype in (torch.float16, torch.bfloat16):
        # under autocast (engine's `args.amp`) the class logits arrive in half precision; the
        # reference's focal loss promotes them to fp32 as well (utils.py:79-104 runs sigmoid /
        # BCE-with-logits, both on autocast's fp32 list).  `.float()` is differentiable: the
        # gradient comes back in the logits' dtype.
        logits = logits.float()
    if logits.dtype != torch.float32:
        raise RuntimeError(f"sigmoid_focal_loss_sums: float32 only, got {logits.dtype}")
    return _FocalSums.apply(logits.contiguous(), target.to(torch.int64).contiguous(), alpha, gamma)


MAX_BOX_LOSS_PAIRS = 3072          # DATR_BOX_LOSS_MAX_PAIRS


class _BoxLossSums(Function):
    """[4, G] per-set sums of (L1, 1 - GIoU, L1 of xy, L1 of wh) over matched pairs
    (csrc/box_loss.hip); gradient to the predicted boxes through rows 0 and 1."""

    @staticmethod
    def forward(ctx, src, tgt, group, G):
        src, tgt, group = src.contiguous(), tgt.contiguous(), group.contiguous()
        sums = torch.empty(4, G, dtype=torch.float32, device=src.device)
        with _native.on_device(src.device):
            rc = _native.lib.datr_box_loss_forward_f32(
                src.data_ptr(), tgt.data_ptr(), group.data_ptr(), src.shape[0], G, sums.data_ptr(),
                _native.current_stream_ptr(src.device))
        _native.check(rc, "box_loss_forward")
        ctx.save_for_backward(src, tgt, group)
        return sums

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_sums):
        src, tgt, group = ctx.saved_tensors
        g = grad_sums.contiguous().float()
        d_src = torch.empty_like(src)
        with _native.on_device(src.device):
            rc = _native.lib.datr_box_loss_backward_f32(
                src.data_ptr(), tgt.data_ptr(), group.data_ptr(), g[0].data_ptr(), g[1].data_ptr(),
                src.shape[0], d_src.data_ptr(), _native.current_stream_ptr(src.device))
        _native.check(rc, "box_loss_backward")
        return d_src, None, None, None


def box_loss_sums(src: torch.Tensor, tgt: torch.Tensor, group: torch.Tensor, G: int) -> torch.Tensor:
    """src, tgt [P, 4] cxcywh fp32 on the device, group [P] int64 in [0, G) -> [4, G] sums of the
    L1 distance, of 1 - GIoU(xyxy(src), xyxy(tgt)) (box_ops.py:40-63, 1e-6 terms included), and
    of the xy / wh halves of the L1, per prediction set.  Rows 2, 3 are for logging: no gradient."""
    if src.dtype in (torch.float16, torch.bfloat16):      # autocast: box losses are fp32 work
        src = src.float()
    tgt = tgt.to(src.dtype)
    assert src.is_cuda and src.dtype == torch.float32 and src.shape == tgt.shape and src.shape[-1] == 4
    assert 0 < src.shape[0] <= MAX_BOX_LOSS_PAIRS and group.dtype == torch.int64
    return _BoxLossSums.apply(src, tgt.detach(), group, int(G))
