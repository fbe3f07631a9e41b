"""A frozen-BN ResNet bottleneck of the NHWC backbone as ONE autograd node on the own kernels.

torchvision's Bottleneck under the reference's FrozenBatchNorm2d
(/root/reference/models/dino/backbone.py:36-72,109-128) is
    y1 = relu(bn1(conv1(x)));  y2 = relu(bn2(conv2(y1)));  y3 = relu(bn3(conv3(y2)) + identity(x))
with identity(x) = x or bn_d(conv_d(x)) (first block of a stage, stride on conv2 and conv_d).
Frozen BN is a per-channel affine, so in the [pixels, channels] view of a channels_last tensor

  forward   y1 = gemm_nt(x, W1 s1; shift1, relu)                       csrc/gemm_f32.hip
            y2 = Winograd / tap-list 3x3 (+ scale2, shift2, relu)      csrc/wino.hip, conv_tap.hip
            y3 = gemm_nt(y2, W3 s3; shift3, residual = identity, relu) -- no affine pass
  backward  (dz3 = dL/dy3 * [y3 > 0] ARRIVES gated when the consumer is the next block of the stage)
            dz2 = gemm_nn(dz3, W3 s3; scale2, gate y2)   -- conv2's frozen-BN + ReLU backward in the epilogue
            dz1 = Winograd data gradient of dz2 with the gate y1 in ITS epilogue
            dx  = gemm_nn(dz1, W1 s1; residual = d identity, gate x)   -- the producer's ReLU backward
            weight gradients: deterministic split-K gemm_tn / Winograd-domain kernel

so a block of a stage's chain runs no element-wise pass at all: the ReLU backward of a block's
OUTPUT is applied by its only consumer, the next block (`gate_in` there, `gated_out` here), which
holds that tensor as its input -- relu's output is zero exactly where its gradient is.
The folded weights W s come from datr_amd.pointwise.fold_frozen_bn (one multi-tensor multiply per
direction for the whole trunk), so the weight gradients returned here are those of the folded weights.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native, gemm
from .wino import wino_conv3x3, wino_filter, wino_filters, wino_wgrad

OWN_BOTTLENECK = os.environ.get("DATR_OWN_BOTTLENECK", "1") != "0"
# below this many output pixels the library GEMMs are faster than the own family by more than the
# element-wise passes cost (4 x 25 x 42 = 4 200: layer4)
MIN_PIXELS = int(os.environ.get("DATR_OWN_BOTTLENECK_MIN_PIXELS", "8192"))


# tests: a list here receives (y1, y2, y3) of every forward (the float64 reference then uses the SAME ReLU
# masks -- a pre-activation within rounding of zero would otherwise gate differently in the two precisions)
_CAPTURE = None


def _rows(x: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] channels_last -> its [N H W, C] view."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


def _like(rows: torch.Tensor, ref_shape, C: int) -> torch.Tensor:
    """[N H W, C] -> the channels_last [N, C, H, W] tensor it is."""
    N, _, H, W = ref_shape
    return rows.view(N, H, W, C).permute(0, 3, 1, 2)


def _relu_gate(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dy * [y > 0] in one pass (csrc/affine_act.hip with a unit scale)."""
    from .pointwise import _ones
    C = dy.shape[1]
    dz = torch.empty_like(dy, memory_format=torch.channels_last)
    with _native.on_device(dy.device):
        rc = _native.lib.datr_affine_act_backward_f32(dy.data_ptr(), y.data_ptr(), _ones(C, dy.device).data_ptr(),
                                                      dy.numel(), C, 1, 1, dz.data_ptr(), 0,
                                                      _native.current_stream_ptr(dy.device))
    _native.check(rc, "affine_act_backward")
    return dz


class _BottleneckFn(Function):
    """See the module docstring.  Inputs: x channels_last; folded 1x1 weights w1s [Cm, Cin], w3s [C4, Cm],
    wds [C4, Cin] or None; conv2's 4-d weight w2 with its frozen scale2 / shift2; the shifts of bn1, bn3
    (+ bn_d added into shift3 by the caller when there is a downsample branch).  stride 1 or 2."""

    @staticmethod
    def forward(ctx, x, w1s, w2, w3s, wds, shift1, scale2, shift2, shift3, shiftd, stride, gate_in, gated_out):
        N, Cin, H, W = x.shape
        Cm, C4 = w1s.shape[0], w3s.shape[0]
        x2 = _rows(x)
        y1r = gemm.gemm_nt(x2, w1s, shift=shift1, relu=True)
        y1 = _like(y1r, x.shape, Cm)
        wt_t = None
        ctx.u2_flipped = None
        if stride == 1:
            # the data gradient's filter comes out of the same launch (kept for backward: the weight does not change
            # in between)
            u2, ctx.u2_flipped = wino_filters(w2, want_flipped=any(ctx.needs_input_grad))
            (y2,) = wino_conv3x3([y1], u2, Cm, shift=shift2, scale=scale2, slope=0.0)
            xs = x
        else:
            from .strided import _workspace
            Ho, Wo = (H + 1) // 2, (W + 1) // 2
            need_t = any(ctx.needs_input_grad[:5])
            wt = torch.empty(9 * Cm * Cm, device=x.device, dtype=torch.float32)
            wt_t = torch.empty(9 * Cm * Cm, device=x.device, dtype=torch.float32) if need_t else None
            sw = w2.stride()
            stream = _native.current_stream_ptr(x.device)
            y2 = torch.empty((N, Cm, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
            ws = _workspace(y1.shape, Cm, x.device)
            with _native.on_device(x.device):
                rc = _native.lib.datr_conv3x3s2_weights_f32(w2.data_ptr(), Cm, Cm, sw[0], sw[1], sw[2], sw[3], wt.data_ptr(),
                                                            0 if wt_t is None else wt_t.data_ptr(), stream)
                _native.check(rc, "conv3x3s2_weights")
                rc = _native.lib.datr_conv3x3s2_forward_nhwc_f32(y1.data_ptr(), wt.data_ptr(), scale2.data_ptr(),
                                                                 shift2.data_ptr(), 0.0, N, H, W, Cm, Cm, y2.data_ptr(),
                                                                 ws.data_ptr(), ws.numel(), stream)
                _native.check(rc, "conv3x3s2_forward")
                xs = x
                if wds is not None:
                    xs = torch.empty((N, Cin, Ho, Wo), device=x.device, dtype=torch.float32,
                                     memory_format=torch.channels_last)
                    rc = _native.lib.datr_even_pixels_nhwc_f32(x.data_ptr(), N, H, W, Cin, xs.data_ptr(), stream)
                    _native.check(rc, "even_pixels")
        y2r = _rows(y2)
        if wds is not None:
            idn = gemm.gemm_nt(_rows(xs), wds, shift=shiftd)
        else:
            idn = x2
        y3r = gemm.gemm_nt(y2r, w3s, shift=shift3, residual=idn, relu=True)
        y3 = _like(y3r, y2.shape, C4)
        if _CAPTURE is not None:
            _CAPTURE.append((y1, y2, y3))
        ctx.save_for_backward(x, xs if wds is not None and stride == 2 else None, y1, y2, y3 if not gated_out else None,
                              w1s, w2, w3s, wds, scale2, wt_t)
        ctx.stride, ctx.gate_in, ctx.gated_out = stride, gate_in, gated_out
        return y3

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, xs, y1, y2, y3, w1s, w2, w3s, wds, scale2, wt_t = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cm, C4 = w1s.shape[0], w3s.shape[0]
        need = ctx.needs_input_grad
        dy = dy.contiguous(memory_format=torch.channels_last)
        dz3 = dy if ctx.gated_out else _relu_gate(dy, y3)
        dz3r, y2r, x2 = _rows(dz3), _rows(y2), _rows(x)
        # conv3: data gradient with conv2's frozen-BN scale and ReLU gate in the epilogue; weight gradient
        dz2r = gemm.gemm_nn(dz3r, w3s, scale=scale2, gate=y2r)
        dz2 = _like(dz2r, y2.shape, Cm)
        dw3s = gemm.gemm_tn(dz3r, y2r) if need[3] else None
        # conv2
        dw2 = None
        if ctx.stride == 1:
            u2f = ctx.u2_flipped if ctx.u2_flipped is not None else wino_filter(w2, True)
            (dz1,) = wino_conv3x3([dz2], u2f, Cm, gates=[y1], gate_slope=0.0)
            if need[2]:
                dw2 = wino_wgrad([y1], [dz2], w2)
        else:
            from .strided import _workspace
            stream = _native.current_stream_ptr(dy.device)
            ws = _workspace(y1.shape, Cm, dy.device)
            d1 = torch.empty_like(y1, memory_format=torch.channels_last)
            with _native.on_device(dy.device):
                rc = _native.lib.datr_conv3x3s2_dgrad_nhwc_f32(dz2.data_ptr(), wt_t.data_ptr(), N, H, W, Cm, Cm,
                                                               d1.data_ptr(), ws.data_ptr(), ws.numel(), stream)
                _native.check(rc, "conv3x3s2_dgrad")
                if need[2]:
                    dw2 = torch.empty_like(w2)
                    s = dw2.stride()
                    rc = _native.lib.datr_conv3x3s2_wgrad_nhwc_f32(y1.data_ptr(), dz2.data_ptr(), N, H, W, Cm, Cm,
                                                                   dw2.data_ptr(), s[0], s[1], s[2], s[3],
                                                                   ws.data_ptr(), ws.numel(), stream)
                    _native.check(rc, "conv3x3s2_wgrad")
            dz1 = _relu_gate(d1, y1)
        dz1r = _rows(dz1)
        dw1s = gemm.gemm_tn(dz1r, x2) if need[1] else None
        # identity branch
        dwds = None
        did = dz3r
        if wds is not None:
            xsr = _rows(xs if xs is not None else x)
            if need[4]:
                dwds = gemm.gemm_tn(dz3r, xsr)
            did = None
            if need[0]:
                dxs = gemm.gemm_nn(dz3r, wds)
                if ctx.stride == 2:
                    full = torch.empty_like(x, memory_format=torch.channels_last)
                    with _native.on_device(dy.device):
                        rc = _native.lib.datr_even_pixels_scatter_nhwc_f32(dxs.data_ptr(), N, H, W, Cin, full.data_ptr(),
                                                                           _native.current_stream_ptr(dy.device))
                    _native.check(rc, "even_pixels_scatter")
                    did = _rows(full)
                else:
                    did = dxs
        dx = None
        if need[0]:
            dxr = gemm.gemm_nn(dz1r, w1s, residual=did, gate=x2 if ctx.gate_in else None)
            dx = _like(dxr, x.shape, Cin)
        return dx, dw1s, dw2, dw3s, dwds, None, None, None, None, None, None, None, None


def applicable(x: torch.Tensor, planes: int, stride: int) -> bool:
    """The node covers channels_last float32 device inputs of bottlenecks whose 3x3 has a multiple of 64
    (stride 1: Winograd kernel) / 128 (stride 2: tap-list kernels) channels, above MIN_PIXELS outputs."""
    if not (OWN_BOTTLENECK and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
            and not torch.is_autocast_enabled()):
        return False
    N, C, H, W = x.shape
    if C % 32 or planes % (64 if stride == 1 else 128) or stride not in (1, 2):
        return False
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    return N * Ho * Wo >= MIN_PIXELS


def bottleneck(x, w1s, w2, w3s, wds, shift1, scale2, shift2, shift3, shiftd, stride, gate_in=False, gated_out=False):
    c = lambda t: None if t is None else t.contiguous()
    return _BottleneckFn.apply(x, w1s, w2, w3s, wds, c(shift1), c(scale2), c(shift2), c(shift3), c(shiftd), stride,
                               bool(gate_in), bool(gated_out))
