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
    with torch.cuda.device(logits.device):
        rc = _native.lib.datr_focal_loss_forward_f32(
            logits.data_ptr(), target.data_ptr(), G, R, C, alpha, gamma, scratch.data_ptr(),
            sums.data_ptr(), _native.current_stream_ptr(logits.device))
    _native.check(rc, "focal_loss_forward")
    return sums


def _focal_backward(logits, target, grad_sums, alpha, gamma):
    G, R, C = logits.shape
    grad = torch.empty_like(logits)
    with torch.cuda.device(logits.device):
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
    if logits.dtype != torch.float32:
        raise RuntimeError(f"sigmoid_focal_loss_sums: float32 only, got {logits.dtype}")
    return _FocalSums.apply(logits.contiguous(), target.to(torch.int64).contiguous(), alpha, gamma)
