"""Small fused element-wise ops backed by libdatr_hip.so."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native


class _AffineAct(Function):
    """y = act(x * scale[c] + shift[c] (+ res)) over NCHW tensors; one HBM pass each way."""

    @staticmethod
    def forward(ctx, x, scale, shift, res, relu):
        x = x.contiguous()
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        if res is not None:
            res = res.contiguous()
        with torch.cuda.device(x.device):
            rc = _native.lib.datr_affine_act_forward_f32(
                x.data_ptr(), 0 if res is None else res.data_ptr(), scale.data_ptr(),
                shift.data_ptr(), x.numel(), C, H * W, int(relu), y.data_ptr(),
                _native.current_stream_ptr(x.device))
        _native.check(rc, "affine_act_forward")
        ctx.relu, ctx.has_res = bool(relu), res is not None
        ctx.save_for_backward(y if relu else None, scale)
        ctx.shape = (C, H * W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        dy = dy.contiguous()
        C, inner = ctx.shape
        dx = torch.empty_like(dy)
        dres = torch.empty_like(dy) if ctx.has_res and ctx.needs_input_grad[3] else None
        with torch.cuda.device(dy.device):
            rc = _native.lib.datr_affine_act_backward_f32(
                dy.data_ptr(), 0 if y is None else y.data_ptr(), scale.data_ptr(), dy.numel(), C,
                inner, int(ctx.relu), dx.data_ptr(), 0 if dres is None else dres.data_ptr(),
                _native.current_stream_ptr(dy.device))
        _native.check(rc, "affine_act_backward")
        return dx, None, None, dres, None


def frozen_bn_act(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                  residual: torch.Tensor = None, relu: bool = True) -> torch.Tensor:
    """Frozen batch-norm as a per-channel affine, optional residual add, optional ReLU.
    Device float32 NCHW tensors take the fused HIP kernel; anything else evaluates the
    reference's own formula (backbone.py:62-72) op by op."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
        return _AffineAct.apply(x, scale.contiguous(), shift.contiguous(), residual, relu)
    y = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y
