"""Winograd F(2x2, 3x3) convolutions on the MFMA units (csrc/wino.hip): the Python side.

`wino_filter` / `wino_conv3x3` wrap the two C-ABI entry points; `conv3x3_bn_relu` is the backbone's
3x3 / stride 1 convolution followed by its frozen batch-norm and ReLU
(/root/reference/models/dino/backbone.py:62-72 around torchvision's Bottleneck.conv2) as one
launch, with the data gradient through the same kernel.  The image-level discriminator's use of the
kernels lives in datr_amd/domain.py.
"""
from __future__ import annotations

import ctypes
import os

import torch
from torch.autograd.function import once_differentiable

# 0 = the backbone's 3x3 convolutions stay on the library (A/B measurements)
OWN_BACKBONE_3X3 = os.environ.get("DATR_OWN_CONV3X3", "1") != "0"
# Widest layer routed through the own kernel: every stride-1 bottleneck of ResNet-50 (64 .. 512 channels).
# Per launch at 4 x 1333x800 (profiles/r04_wino_s32.txt, conv + frozen BN + ReLU in one launch): 64 ch 139 us,
# 128 ch 132 us, 256 ch 144 us, 512 ch 167 us; the library's convolution alone (without its affine + ReLU pass)
# ran 216 / 191 / 184 / 176 us on these layers (profiles/r02_wino.md).
OWN_BACKBONE_3X3_MAX_CH = int(os.environ.get("DATR_OWN_CONV3X3_MAX_CH", "512"))


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous(memory_format=torch.channels_last)


def wino_filter(w: torch.Tensor, data_gradient: bool = False) -> torch.Tensor:
    """Winograd-domain filter G g G^T of a [Cout, Cin, 3, 3] weight in the layout csrc/wino.hip reads
    ([16][Cin/8][Cout][8], the 8 channels of a block in MFMA-lane order); data_gradient=True gives the filter of the transposed convolution
    (channels swapped, taps mirrored)."""
    from . import _native
    co, ci = w.shape[:2]
    s = w.stride()
    if data_gradient:
        co, ci, s = ci, co, (s[1], s[0], s[2], s[3])
    u = torch.empty(16 * ci * co, device=w.device, dtype=torch.float32)
    with _native.on_device(w.device):
        rc = _native.lib.datr_wino_weights_f32(w.data_ptr(), co, ci, s[0], s[1], s[2], s[3],
                                               1 if data_gradient else 0, u.data_ptr(),
                                               _native.current_stream_ptr(w.device))
    _native.check(rc, "wino_weights")
    return u


def wino_filters(w: torch.Tensor, want_flipped: bool = True):
    """(forward filter, data-gradient filter or None): one launch for both where the channel counts admit it."""
    if want_flipped:
        pair = wino_filter_pair(w)
        if pair is not None:
            return pair
    return wino_filter(w), None


def wino_filter_pair(w: torch.Tensor):
    """(wino_filter(w), wino_filter(w, data_gradient=True)) from ONE launch -- the mirrored filter's transform is a
    permutation of the plain one's -- or None when the channel counts do not admit both (multiples of 64)."""
    from . import _native
    co, ci = w.shape[:2]
    if co % 64 or ci % 64:
        return None
    s = w.stride()
    u = torch.empty(16 * ci * co, device=w.device, dtype=torch.float32)
    uf = torch.empty(16 * ci * co, device=w.device, dtype=torch.float32)
    with _native.on_device(w.device):
        rc = _native.lib.datr_wino_weights_pair_f32(w.data_ptr(), co, ci, s[0], s[1], s[2], s[3], u.data_ptr(),
                                                    uf.data_ptr(), _native.current_stream_ptr(w.device))
    _native.check(rc, "wino_weights_pair")
    return u, uf


def wino_conv3x3(xs, u: torch.Tensor, cout: int, shift=None, scale=None, slope: float = 1.0, gates=None,
                 gate_slope: float = 1.0, out_scale: float = 1.0):
    """3x3 / stride 1 / pad 1 convolution of every level in `xs` (channels_last [N, Cin, H, W] device
    tensors sharing the filter `u` from wino_filter) in ONE launch of csrc/wino.hip:
    out = out_scale * gate(lrelu_slope(scale * conv + shift)); returns channels_last tensors."""
    from . import _native
    assert 1 <= len(xs) <= 4
    N, cin = xs[0].shape[:2]
    ys = [torch.empty((N, cout) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32,
                      memory_format=torch.channels_last) for x in xs]
    levels = (_native.WinoLevel * len(xs))()
    for i, (x, y) in enumerate(zip(xs, ys)):
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[:2] == (N, cin)
        assert x.is_contiguous(memory_format=torch.channels_last)
        g = None if gates is None else gates[i]
        if g is not None:
            assert g.shape == y.shape and g.is_contiguous(memory_format=torch.channels_last)
        levels[i] = _native.WinoLevel(x.data_ptr(), y.data_ptr(), 0 if g is None else g.data_ptr(),
                                      x.shape[2], x.shape[3])
    with _native.on_device(xs[0].device):
        rc = _native.lib.datr_conv3x3_wino_nhwc_f32(
            ctypes.addressof(levels), len(xs), N, cin, cout, u.data_ptr(),
            0 if scale is None else scale.data_ptr(), 0 if shift is None else shift.data_ptr(),
            slope, gate_slope, out_scale, _native.current_stream_ptr(xs[0].device))
    _native.check(rc, "conv3x3_wino_nhwc")
    return ys


def wino_wgrad(xs, dys, weight: torch.Tensor) -> torch.Tensor:
    """Weight gradient of the 3x3 / stride 1 / pad 1 convolution with filter `weight` ([Cout, Cin, 3, 3],
    any strides) summed over the levels: xs[i] the layer's input, dys[i] the gradient of its output,
    channels_last [N, C, H, W] device tensors.  ONE launch of csrc/wino_wgrad.hip plus the fold; the
    result has weight's shape and strides.  None when the channel counts are not multiples of 64."""
    from . import _native
    co, ci = weight.shape[:2]
    if co % 64 or ci % 64 or tuple(weight.shape[2:]) != (3, 3):
        return None
    assert 1 <= len(xs) == len(dys) <= 4
    N = xs[0].shape[0]
    levels = (_native.WinoWgradLevel * len(xs))()
    for i, (x, dy) in enumerate(zip(xs, dys)):
        assert x.is_cuda and x.dtype == dy.dtype == torch.float32
        assert x.shape == (N, ci) + tuple(dy.shape[2:]) and dy.shape[:2] == (N, co)
        assert x.is_contiguous(memory_format=torch.channels_last) and dy.is_contiguous(memory_format=torch.channels_last)
        levels[i] = _native.WinoWgradLevel(x.data_ptr(), dy.data_ptr(), x.shape[2], x.shape[3])
    floats = _native.lib.datr_wino_wgrad_partial_floats(ctypes.addressof(levels), len(xs), N, ci, co)
    if floats < 0:
        return None
    partial = torch.empty(floats, device=weight.device, dtype=torch.float32)
    dw = torch.empty_like(weight)                                   # preserves the strides
    s = dw.stride()
    with _native.on_device(weight.device):
        rc = _native.lib.datr_conv3x3_wino_wgrad_nhwc_f32(ctypes.addressof(levels), len(xs), N, ci, co,
                                                          partial.data_ptr(), dw.data_ptr(), s[0], s[1], s[2], s[3],
                                                          _native.current_stream_ptr(weight.device))
    _native.check(rc, "conv3x3_wino_wgrad_nhwc")
    return dw


class _Conv3x3BnRelu(torch.autograd.Function):
    """relu(conv3x3(x, w) * scale + shift), NHWC, frozen scale / shift (buffers, no gradient)."""

    @staticmethod
    def forward(ctx, x, w, scale, shift):
        x = _nhwc(x)
        (y,) = wino_conv3x3([x], wino_filter(w), w.shape[0], shift=shift, scale=scale, slope=0.0)
        ctx.save_for_backward(x, w, y, scale)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from . import _native
        x, w, y, scale = ctx.saved_tensors
        dy = _nhwc(dy)
        # dz = dy * [y > 0] * scale: the frozen-BN + ReLU backward pass (csrc/affine_act.hip)
        dz = torch.empty_like(dy, memory_format=torch.channels_last)
        with _native.on_device(dy.device):
            rc = _native.lib.datr_affine_act_backward_f32(
                dy.data_ptr(), y.data_ptr(), scale.data_ptr(), dy.numel(), dy.shape[1], 1, 1, dz.data_ptr(), 0,
                _native.current_stream_ptr(dy.device))
        _native.check(rc, "affine_act_backward")
        dx = dw = None
        if ctx.needs_input_grad[0]:
            (dx,) = wino_conv3x3([dz], wino_filter(w, True), w.shape[1])
        if ctx.needs_input_grad[1]:
            dw = wino_wgrad([x], [dz], w)
            if dw is None:
                _, dw, _ = torch.ops.aten.convolution_backward(dz, x, w, None, [1, 1], [1, 1], [1, 1], False,
                                                               [0, 0], 1, [False, True, False])
        return dx, dw, None, None


class _Conv3x3OwnWgrad(torch.autograd.Function):
    """conv2d(x, w, stride 1, padding 1) with the library's forward and data gradient and the
    Winograd-domain weight gradient (csrc/wino_wgrad.hip): bottlenecks
    wider than DATR_OWN_CONV3X3_MAX_CH (none in ResNet-50 at the default of 512: a fallback)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.conv2d(x, w, padding=1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _nhwc(dy)
        dx = dw = None
        conv_bwd = torch.ops.aten.convolution_backward
        if ctx.needs_input_grad[0]:
            dx, _, _ = conv_bwd(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
        if ctx.needs_input_grad[1]:
            dw = wino_wgrad([x], [dy], w)
            if dw is None:
                _, dw, _ = conv_bwd(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
        return dx, dw


def conv3x3_own_wgrad(x: torch.Tensor, w: torch.Tensor):
    """Library convolution whose weight gradient is the own kernel, or None when that kernel does not
    apply (the caller then calls the module)."""
    co, ci = w.shape[:2]
    if not (OWN_BACKBONE_3X3 and w.requires_grad and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and w.shape[2:] == (3, 3)
            and ci % 64 == 0 and co % 64 == 0 and not torch.is_autocast_enabled()):
        return None
    return _Conv3x3OwnWgrad.apply(x, w)


def conv3x3_bn_relu(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor):
    """relu(frozen_bn(conv2d(x, w, stride 1, padding 1))) in one Winograd/MFMA launch, or None when
    the shapes / dtype / device / layout are not the kernel's -- it is an NHWC kernel, an NCHW
    backbone stays on the library -- (the caller then takes the library path)."""
    co, ci = w.shape[:2]
    if not (OWN_BACKBONE_3X3 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and w.shape[2:] == (3, 3) and ci % 8 == 0 and co % 64 == 0
            and ci <= OWN_BACKBONE_3X3_MAX_CH and (not w.requires_grad or ci % 64 == 0)
            and not torch.is_autocast_enabled()):
        return None
    return _Conv3x3BnRelu.apply(x, w, scale.contiguous(), shift.contiguous())
