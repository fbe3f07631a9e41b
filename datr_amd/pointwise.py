"""1x1 convolutions of the NHWC backbone as GEMMs on the [pixels, channels] view.

A 1x1 / stride 1 convolution of a torch.channels_last tensor IS `x2d @ W^T` on the zero-copy
[N*H*W, C] view.  MIOpen runs these layers with implicit-GEMM convolution kernels at 66-70 % matrix
pipe utilisation; the same arithmetic through hipBLASLt (TunableOp-selected kernels) is 10-15 %
faster forward and for the data gradient (tools/probes/conv1x1_gemm_probe.py), and its bias / ReLU
epilogue takes the frozen batch norm that follows conv1 of every ResNet bottleneck
(/root/reference/models/dino/backbone.py:62-72 around torchvision's Bottleneck):
    relu(bn(conv1(x))) = relu(x2d @ (W * scale)^T + shift)          -- ONE GEMM, no affine pass.
The weight gradient is the own split-K kernel (datr_amd.gemm.gemm_tn) on the large maps (a library
GEMM whose reduction axis is 10^4 .. 10^5 long loses by up to 2.5x) and a library GEMM on the small ones.

`fold_frozen_bn` multiplies the weights of many layers by their frozen scales in ONE multi-tensor
launch each way (folded weights of frozen layers are cached).
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native

# 0 = the library convolutions (A/B measurements)
GEMM_1X1 = os.environ.get("DATR_CONV1X1_GEMM", "1") != "0"
# weight gradient as a LIBRARY GEMM below this many pixels (measured: at 4 x 25 x 42 = 4 200 pixels it
# is 65-75 us against 79-90 us for the own split-K kernel; at 16 800 and more the own kernel wins 1.6-4x)
WGRAD_GEMM_MAX_PIXELS = int(os.environ.get("DATR_CONV1X1_WGRAD_GEMM_MAX_PIXELS", "5000"))

_ONES = {}


def _ones(c: int, device) -> torch.Tensor:
    key = (c, device)
    if key not in _ONES:
        _ONES[key] = torch.ones(c, device=device, dtype=torch.float32)
    return _ONES[key]


class _Conv1x1(Function):
    """y = act(conv1x1(x, w) + b) for a channels_last x, as a GEMM on the [pixels, C] view."""

    @staticmethod
    def forward(ctx, x, w2, bias, relu):
        N, C, H, W = x.shape
        co = w2.shape[0]
        x2 = x.permute(0, 2, 3, 1).reshape(-1, C)                     # zero-copy for channels_last
        from . import gemm
        if gemm.own_big(x2, w2):          # no valid hipBLASLt selections: the own NT form (datr_amd.gemm.BACKEND)
            y2 = gemm.gemm_nt(x2, w2, shift=None if bias is None else bias.contiguous(), relu=bool(relu))
        elif relu:
            y2 = torch._addmm_activation(bias, x2, w2.t(), use_gelu=False)
        elif bias is not None:
            y2 = torch.addmm(bias, x2, w2.t())
        else:
            y2 = x2.mm(w2.t())
        y = y2.view(N, H, W, co).permute(0, 3, 1, 2)                  # a channels_last [N, co, H, W]
        ctx.save_for_backward(x, w2, y if relu else None)
        ctx.relu, ctx.has_bias = bool(relu), bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w2, y = ctx.saved_tensors
        N, C, H, W = x.shape
        co = w2.shape[0]
        need = ctx.needs_input_grad
        dy = dy.contiguous(memory_format=torch.channels_last)
        if ctx.relu:                                                  # dz = dy * [y > 0], one pass
            dz = torch.empty_like(dy, memory_format=torch.channels_last)
            with _native.on_device(dy.device):
                rc = _native.lib.datr_affine_act_backward_f32(
                    dy.data_ptr(), y.data_ptr(), _ones(co, dy.device).data_ptr(), dy.numel(), co, 1, 1,
                    dz.data_ptr(), 0, _native.current_stream_ptr(dy.device))
            _native.check(rc, "affine_act_backward")
        else:
            dz = dy
        dz2 = dz.permute(0, 2, 3, 1).reshape(-1, co)
        dx = dw = db = None
        if need[0]:
            from . import gemm
            dx = (gemm.gemm_nn(dz2, w2) if gemm.own_big(dz2, w2) else dz2.mm(w2)).view(N, H, W, C).permute(0, 3, 1, 2)
        if need[1]:
            x2 = x.permute(0, 2, 3, 1).reshape(-1, C)
            if N * H * W <= WGRAD_GEMM_MAX_PIXELS:
                dw = dz2.t().mm(x2)
            else:
                # deterministic split-K product over the pixels on the own MFMA kernel (csrc/gemm_f32.hip):
                # a library GEMM whose reduction axis is 10^4 .. 10^5 long loses by up to 2.5x
                from . import gemm
                dw = gemm.gemm_tn(dz2, x2)
        if ctx.has_bias and need[2]:
            from .fused import column_sums
            db = column_sums(dz2)
        return dx, dw, db, None


def conv1x1(x: torch.Tensor, weight: torch.Tensor, bias=None, relu: bool = False):
    """act(F.conv2d(x, weight[Cout, Cin, 1, 1] or [Cout, Cin], bias)) through the GEMM path, or None
    when x is not a device float32 channels_last tensor (the caller then takes the library path).
    `relu` needs a bias (the epilogue is bias + ReLU)."""
    if not (GEMM_1X1 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and not torch.is_autocast_enabled()
            and x.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0 and (bias is not None or not relu)):
        return None
    w2 = weight.reshape(weight.shape[0], -1)
    if w2.shape[1] != x.shape[1]:
        return None
    return _Conv1x1.apply(x, w2, bias, relu)


class _FoldMany(Function):
    """[w_i * s_i[:, None] for i] with one multi-tensor launch forward and one backward."""

    @staticmethod
    def forward(ctx, n, *args):
        ws, ss = args[:n], args[n:]
        ctx.save_for_backward(*ss)
        ctx.set_materialize_grads(False)         # a folded weight nobody used gets no gradient
        return tuple(torch._foreach_mul(list(ws), [s.view(-1, 1) for s in ss]))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        ss = ctx.saved_tensors
        gs = [g if g is not None else None for g in gs]
        live = [i for i, g in enumerate(gs) if g is not None]
        out = [None] * len(gs)
        if live:
            prod = torch._foreach_mul([gs[i] for i in live], [ss[i].view(-1, 1) for i in live])
            for i, p in zip(live, prod):
                out[i] = p
        return (None, *out, *([None] * len(ss)))


_FROZEN_FOLDS = {}


def fold_frozen_bn(pairs):
    """pairs: [(conv weight [Cout, Cin, 1, 1], frozen scale [Cout])] -> folded [Cout, Cin] weights.
    Weights that do not require a gradient are folded once and cached (keyed by storage + version of
    both tensors); the others go through one multi-tensor multiply per direction."""
    out = [None] * len(pairs)
    train = []
    for i, (w, s) in enumerate(pairs):
        if w.requires_grad and torch.is_grad_enabled():
            train.append(i)
            continue
        key = (w.data_ptr(), w._version, s.data_ptr(), s._version)
        hit = _FROZEN_FOLDS.get(key)
        if hit is None:
            if len(_FROZEN_FOLDS) > 256:
                _FROZEN_FOLDS.clear()
            with torch.no_grad():
                hit = _FROZEN_FOLDS[key] = ((w.reshape(w.shape[0], -1) * s.view(-1, 1)).contiguous(), w, s)
        out[i] = hit[0]
    if train:
        folded = _FoldMany.apply(len(train), *[pairs[i][0].reshape(pairs[i][0].shape[0], -1) for i in train],
                                 *[pairs[i][1] for i in train])
        for i, f in zip(train, folded):
            out[i] = f
    return out
