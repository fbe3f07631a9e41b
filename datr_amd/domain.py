"""Domain-adaptation pieces of DATR: gradient reversal, the image-level discriminator and
class-wise query prototypes.

Mirror of /root/reference/models/dino/DA_utils.py (`decompose_features` :5-31, `GradReverse`
:33-43, `FCDiscriminator_img` :61-79, `get_prototype_class_wise` :82-120).  The prototype
extraction computes the same class-wise means with one [C, B*N] x [B*N, 256] product instead
of materialising the reference's [B*N, C, 256] masked copy (:96-108); sums are reassociated,
so values agree to fp32 rounding, the argmax class map and counts are exact.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd.function import once_differentiable

from .wino import _nhwc, wino_conv3x3, wino_filter, wino_filters, wino_wgrad


def _c1_levels(xs, ys, dxs=None):
    from . import _native
    levels = (_native.C1Level * len(xs))()
    for i, x in enumerate(xs):
        levels[i] = _native.C1Level(x.data_ptr(), ys[i].data_ptr(), 0 if dxs is None else dxs[i].data_ptr(),
                                    x.shape[2], x.shape[3])
    return levels


def classifier_forward(acts, weight, bias):
    """[conv2d(a, weight, bias, padding=1) for a in acts] for a one-output-channel 3x3 filter on
    channels_last levels [N, C, H, W] (csrc/conv_cout1.hip: a stream over the activations, all
    levels per call).  -> list of [N, 1, H, W]."""
    import ctypes
    from . import _native
    n = acts[0].shape[0]
    outs = [torch.empty((n, 1, a.shape[2], a.shape[3]), dtype=a.dtype, device=a.device) for a in acts]
    w = weight.contiguous()
    with _native.on_device(acts[0].device):
        levels = _c1_levels(acts, outs)
        rc = _native.lib.datr_conv3x3_cout1_forward_f32(ctypes.addressof(levels), len(acts), n, acts[0].shape[1],
                                                        w.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                                        _native.current_stream_ptr(acts[0].device))
    _native.check(rc, "conv3x3_cout1_forward")
    return outs


def classifier_backward(acts, douts, weight, slope):
    """-> (dz per level = LeakyReLU-gated data gradient, channels_last; dW [1, C, 3, 3]; db [1])."""
    import ctypes
    from . import _native
    n = acts[0].shape[0]
    douts = [d.contiguous() for d in douts]
    dzs = [torch.empty_like(a) for a in acts]                     # channels_last like the activations
    w = weight.contiguous()
    dw, db = torch.empty_like(w), torch.empty(1, dtype=w.dtype, device=w.device)
    levels = _c1_levels(acts, douts, dzs)
    floats = int(_native.lib.datr_conv3x3_cout1_partial_floats(ctypes.addressof(levels), len(acts), n))
    partial = torch.empty(max(floats, 1), dtype=torch.float32, device=w.device)
    with _native.on_device(w.device):
        rc = _native.lib.datr_conv3x3_cout1_backward_f32(ctypes.addressof(levels), len(acts), n, acts[0].shape[1], w.data_ptr(),
                                                         ctypes.c_float(slope), dw.data_ptr(), db.data_ptr(),
                                                         partial.data_ptr(), _native.current_stream_ptr(w.device))
    _native.check(rc, "conv3x3_cout1_backward")
    return dzs, dw, db

# own Winograd-on-MFMA path for the discriminator's 3x3 convolutions (csrc/wino.hip); 0 = the
# library convolutions under autograd (A/B measurements)
OWN_D_IMG = os.environ.get("DATR_OWN_D_IMG", "1") != "0"
OWN_D_IMG_WGRAD = os.environ.get("DATR_OWN_D_IMG_WGRAD", "1") != "0"     # weight gradients in the Winograd domain too
OWN_CLASSIFIER = os.environ.get("DATR_OWN_CLASSIFIER", "1") != "0"        # the 128 -> 1 classifier (csrc/conv_cout1.hip)


def decompose_features(srcs, masks, poss):
    """Split every level's batch into (source half, all, target half)."""
    half = srcs[0].shape[0] // 2
    src = ([s[:half] for s in srcs], [m[:half] for m in masks], [p[:half] for p in poss])
    tgt = ([s[half:] for s in srcs], [m[half:] for m in masks], [p[half:] for p in poss])
    return (*src, srcs, masks, poss, *tgt)


class GradReverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.neg()


def grad_reverse(x):
    return GradReverse.apply(x)


class _DImgPyramid(torch.autograd.Function):
    """Gradient reversal + FCDiscriminator_img on all pyramid levels (DA_utils.py:33-79 as called
    from dino.py:351-359).  Forward: three Winograd/MFMA launches (bias + LeakyReLU in the epilogue,
    all levels per launch) and the 128 -> 1 classifier.  Backward: per layer the data gradient is the
    same kernel on the transposed filter with the previous layer's LeakyReLU gate -- and, for the first
    layer, the reversal's minus sign -- in its epilogue; the weight gradient of a layer is one launch
    of csrc/wino_wgrad.hip over all levels (the same Winograd domain: sum over tiles of
    (A dY A^T) o (B^T x B), folded by G^T . G), the bias gradient a column sum.  The 128 -> 1 classifier
    is a streaming kernel of its own (csrc/conv_cout1.hip), forward and backward."""

    SLOPE = 0.2

    @staticmethod
    def forward(ctx, w1, b1, w2, b2, w3, b3, wc, bc, *xs):
        xs = [_nhwc(x) for x in xs]
        # forward and data-gradient filters of a layer from one launch (kept for backward)
        bwd = any(ctx.needs_input_grad)
        (u1, f1), (u2, f2), (u3, f3) = (wino_filters(w, bwd) for w in (w1, w2, w3))
        ctx.flipped = (f1, f2, f3)
        a1 = wino_conv3x3(xs, u1, w1.shape[0], shift=b1, slope=_DImgPyramid.SLOPE)
        a2 = wino_conv3x3(a1, u2, w2.shape[0], shift=b2, slope=_DImgPyramid.SLOPE)
        a3 = wino_conv3x3(a2, u3, w3.shape[0], shift=b3, slope=_DImgPyramid.SLOPE)
        ctx.own_classifier = OWN_CLASSIFIER and wc.shape[:2] == (1, 128) and all(a.shape[1] == 128 for a in a3)
        outs = classifier_forward(a3, wc, bc) if ctx.own_classifier else [F.conv2d(a, wc, bc, padding=1) for a in a3]
        ctx.save_for_backward(w1, w2, w3, wc, *xs, *a1, *a2, *a3)
        ctx.levels = len(xs)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *douts):
        n = ctx.levels
        w1, w2, w3, wc = ctx.saved_tensors[:4]
        rest = ctx.saved_tensors[4:]
        xs, a1, a2, a3 = (rest[i * n:(i + 1) * n] for i in range(4))
        slope = _DImgPyramid.SLOPE
        conv_bwd = torch.ops.aten.convolution_backward

        def wgrad(dzs, ins, w):
            """sum over levels of (dW, db): the Winograd-domain weight gradient, all levels in one
            launch (csrc/wino_wgrad.hip; the library's weight-gradient convolution per level for
            channel counts it does not take); the bias
            gradient is the column sum of dz viewed [pixels, C] (NHWC), a deterministic two-stage own
            kernel (csrc/ffn.hip) instead of ATen's 64-workgroup reduction"""
            from .fused import column_sums
            dw = wino_wgrad(ins, dzs, w) if OWN_D_IMG_WGRAD else None
            db = None
            for dz, a in zip(dzs, ins):
                gb = column_sums(dz.permute(0, 2, 3, 1).reshape(-1, dz.shape[1]))
                db = gb if db is None else db.add_(gb)
            if dw is None:                                   # channel counts the kernel does not take
                for dz, a in zip(dzs, ins):
                    _, gw, _ = conv_bwd(dz, a, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                        [False, True, False])
                    dw = gw if dw is None else dw.add_(gw)
            return dw, db

        # classifier (128 -> 1): own streaming kernels (csrc/conv_cout1.hip), the LeakyReLU gate of
        # conv3 applied in the data gradient; DATR_OWN_CLASSIFIER=0: library kernels, gated by hand
        dz3, dwc, dbc = [], None, None
        if ctx.own_classifier:
            dz3, dwc, dbc = classifier_backward(a3, douts, wc, slope)
        for do, a in (() if ctx.own_classifier else zip(douts, a3)):
            da, gw, gb = conv_bwd(do.contiguous(), a, wc, [1], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                  [True, True, True])
            dz3.append(_nhwc(torch.where(a > 0, da, da * slope)))
            dwc = gw if dwc is None else dwc.add_(gw)
            dbc = gb if dbc is None else dbc.add_(gb)
        dw3, db3 = wgrad(dz3, a2, w3)
        f1, f2, f3 = (f if f is not None else wino_filter(w, True) for f, w in zip(ctx.flipped, (w1, w2, w3)))
        dz2 = wino_conv3x3(dz3, f3, w3.shape[1], gates=a2, gate_slope=slope)
        dw2, db2 = wgrad(dz2, a1, w2)
        dz1 = wino_conv3x3(dz2, f2, w2.shape[1], gates=a1, gate_slope=slope)
        dw1, db1 = wgrad(dz1, xs, w1)
        dxs = [None] * n
        if any(ctx.needs_input_grad[8:]):
            dxs = wino_conv3x3(dz1, f1, w1.shape[1], out_scale=-1.0)   # GRL
        return (dw1, db1, dw2, db2, dw3, db3, dwc, dbc, *dxs)


class FCDiscriminator_img(nn.Module):
    """3x3 convs 256->256->128->128->1 with LeakyReLU(0.2) between (per-pixel domain logit)."""

    def __init__(self, num_classes, ndf1=256, ndf2=128):
        super().__init__()
        self.conv1 = nn.Conv2d(num_classes, ndf1, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(ndf1, ndf2, kernel_size=3, padding=1)
        self.conv3 = nn.Conv2d(ndf2, ndf2, kernel_size=3, padding=1)
        self.classifier = nn.Conv2d(ndf2, 1, kernel_size=3, padding=1)
        self.leaky_relu = nn.LeakyReLU(negative_slope=0.2, inplace=True)

    def forward(self, x):
        x = self.leaky_relu(self.conv1(x))
        x = self.leaky_relu(self.conv2(x))
        x = self.leaky_relu(self.conv3(x))
        return self.classifier(x)

    def reversed_pyramid(self, srcs):
        """[self(grad_reverse(s)) for s in srcs] -- the call of dino.py:351-359.  Device float32
        levels go through the fused Winograd/MFMA path (_DImgPyramid, all levels per launch)."""
        srcs = list(srcs)
        convs = (self.conv1, self.conv2, self.conv3)
        own = (OWN_D_IMG and 1 <= len(srcs) <= 4 and all(s.is_cuda and s.dtype == torch.float32 for s in srcs)
               and all(c.in_channels % 8 == 0 and c.out_channels % 64 == 0 for c in convs)
               and not torch.is_autocast_enabled())
        if not own:
            return [self(grad_reverse(s)) for s in srcs]
        return list(_DImgPyramid.apply(self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                       self.conv3.weight, self.conv3.bias, self.classifier.weight,
                                       self.classifier.bias, *srcs))


OWN_PROTOTYPES = __import__("os").environ.get("DATR_OWN_PROTOTYPES", "1") != "0"      # A/B switch


class _ClassPrototypes(torch.autograd.Function):
    """(prototypes, present, new_global, new_amount, onehot) of get_prototype_class_wise from the labels on:
    one launch forward, one backward (only the prototypes carry a gradient, to the features)."""

    @staticmethod
    def forward(ctx, feats, labels, global_proto, amount, K):
        from . import _native
        R, C = feats.shape
        dev = feats.device
        proto = torch.empty(K, C, device=dev, dtype=torch.float32)
        present = torch.empty(K, device=dev, dtype=torch.float32)
        new_global = torch.empty(K, C, device=dev, dtype=torch.float32)
        new_amount = torch.empty(K, device=dev, dtype=torch.float32)
        onehot = torch.empty(R, K, device=dev, dtype=torch.float32)
        with _native.on_device(dev):
            rc = _native.lib.datr_class_prototypes_forward_f32(
                feats.data_ptr(), labels.data_ptr(), global_proto.data_ptr(), amount.data_ptr(), R, C, K,
                proto.data_ptr(), present.data_ptr(), new_global.data_ptr(), new_amount.data_ptr(), onehot.data_ptr(),
                _native.current_stream_ptr(dev))
        _native.check(rc, "class_prototypes_forward")
        ctx.save_for_backward(labels, new_amount - amount)
        ctx.shape = (R, C, K)
        ctx.mark_non_differentiable(present, new_global, new_amount, onehot)
        return proto, present, new_global, new_amount, onehot

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_proto, *_):
        from . import _native
        labels, count = ctx.saved_tensors
        R, C, K = ctx.shape
        d_proto = d_proto.contiguous()
        d_feats = torch.empty(R, C, device=d_proto.device, dtype=torch.float32)
        with _native.on_device(d_proto.device):
            rc = _native.lib.datr_class_prototypes_backward_f32(d_proto.data_ptr(), labels.data_ptr(), count.data_ptr(),
                                                                R, C, K, d_feats.data_ptr(),
                                                                _native.current_stream_ptr(d_proto.device))
        _native.check(rc, "class_prototypes_backward")
        return d_feats, None, None, None, None


def get_prototype_class_wise(object_query_last_layer, outputs_class, num_classes,
                             global_proto=None, global_amount=None):
    """Class-wise mean of the last-layer queries, with classes assigned by argmax of the
    predicted scores; also advances the running (count-weighted) global prototypes.

    Returns (prototypes [C,256], present [C] in {0,1}, new_global [C,256] detached,
             new_amount [C], onehot [B*N, C])."""
    B, N, C = object_query_last_layer.shape
    labels = torch.argmax(outputs_class.sigmoid(), dim=2).reshape(B * N, 1)
    feats = object_query_last_layer.reshape(B * N, C)
    if OWN_PROTOTYPES and feats.is_cuda and feats.dtype == torch.float32 and C % 64 == 0 and global_proto is not None \
            and global_proto.dtype == torch.float32 and not global_proto.requires_grad \
            and not torch.is_autocast_enabled():
        # everything after the labels as one launch (csrc/prototypes.hip)
        return _ClassPrototypes.apply(feats if feats.is_contiguous() else feats.contiguous(), labels.view(-1),
                                      global_proto.contiguous(), global_amount.contiguous().float(), num_classes)
    onehot = torch.zeros(B * N, num_classes, device=feats.device)
    onehot.scatter_(dim=1, index=labels, value=1)
    count = onehot.sum(0)                                          # [num_classes]
    present = torch.where(count != 0, torch.ones_like(count), count)
    denom = torch.where(count == 0, torch.ones_like(count), count)
    prototypes = (onehot.t() @ feats) / denom[:, None]
    weight = count / (count + global_amount)
    weight = torch.where(count == 0, torch.zeros_like(weight), weight)[:, None]
    new_global = (global_proto * (1 - weight) + prototypes * weight).detach()
    return prototypes, present, new_global, global_amount + count, onehot
