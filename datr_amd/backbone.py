"""ResNet-50 backbone with frozen batch-norm, padding masks and sine position encoding.

Mirror of /root/reference/models/dino/backbone.py (`FrozenBatchNorm2d` :36-72, `BackboneBase`
:75-106, `Backbone` :109-128, `Joiner` :131-144, `build_backbone` :147-219) and of
/root/reference/models/dino/position_encoding.py (`PositionEmbeddingSineHW` :62-108).

The ResNet-50 arithmetic itself is NOT in the reference tree: it comes from torchvision
(`torchvision.models.resnet50`, requirements.txt:5, un-vendored).  `ResNet50Body` restates the
public v1.5 architecture (stride on the 3x3 conv of each bottleneck) with torchvision's
parameter names so that DATR / DINO checkpoints load unchanged
(`backbone.0.body.layerK.J.{conv,bn}N.*`).  Parity for it is unpinned by any
reference test; tests/test_backbone_cpu.py / test_backbone_gpu.py pin names, shapes, the trainable split and
the arithmetic against oracle/resnet_ref.py (a plain-nn restatement that shares no code with this file).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import nn

from .fused import fan_out, frozen_bn_act
from . import strided
from .wino import conv3x3_bn_relu, conv3x3_own_wgrad
from . import bottleneck, pointwise
from .nested import NestedTensor


class FrozenBatchNorm2d(nn.Module):
    """y = x * w/sqrt(var+eps) + (b - mean * w/sqrt(var+eps)); statistics and affine are
    buffers (never trained), eps = 1e-5 inside the rsqrt (backbone.py:62-72)."""

    def __init__(self, n: int):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)   # torchvision BN checkpoints
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    @torch.no_grad()
    def scale_shift(self):
        """(scale, shift) of the equivalent per-channel affine; cached until a buffer changes
        (load_state_dict, .to(device)) so a training step does not recompute 53 x 4 tiny ops."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b.data_ptr(), b._version) for b in bufs)
        cache = getattr(self, "_affine_cache", None)
        if cache is None or cache[0] != key:
            scale = self.weight * (self.running_var + 1e-5).rsqrt()
            cache = (key, scale, self.bias - self.running_mean * scale)
            object.__setattr__(self, "_affine_cache", cache)
        return cache[1], cache[2]

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


_CAPTURE = None


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, norm_layer, downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                norm_layer(planes * 4))

    # set on every block but the last of a stage (ResNet50Body): the block then returns its output as TWO
    # handles (for the next block's conv1 and identity branch) whose gradients the frozen-BN backward
    # kernel adds on the fly; the next block takes the pair apart
    pair_out = False

    def forward(self, x):
        x_id = x
        if isinstance(x, tuple):
            x, x_id = x
        if not isinstance(self.bn1, FrozenBatchNorm2d):       # foreign norm layer: plain path
            identity = x if self.downsample is None else self.downsample(x)
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.relu(self.bn2(self.conv2(out)))
            return self.relu(self.bn3(self.conv3(out)) + identity)
        # frozen BN = per-channel affine: fold it with the ReLU / residual that follows into one
        # pass over the activation (datr_amd.fused.frozen_bn_act); in the NHWC layout the 1x1
        # convolutions are GEMMs on the [pixels, channels] view (datr_amd.pointwise): conv1 + bn1 +
        # ReLU is ONE GEMM with the scale folded into the weight and shift / ReLU in the epilogue
        nhwc = (pointwise.GEMM_1X1 and x.is_cuda and x.dtype == torch.float32
                and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())
        folded = getattr(self, "_folded", None) if nhwc else None
        ds_conv = None if self.downsample is None else self.downsample[0]
        if self.downsample is None:
            identity = x_id
        else:
            identity = None
            if nhwc and folded is not None and folded[1] is not None:
                shift_ds = self.downsample[1].scale_shift()[1]
                if ds_conv.stride == (1, 1):
                    identity = pointwise.conv1x1(x_id, folded[1], shift_ds)
                elif ds_conv.stride == (2, 2):           # even pixels gathered, then the same GEMM
                    identity = strided.conv1x1_s2(x_id, folded[1], shift_ds)
            if identity is None:
                identity = frozen_bn_act(ds_conv(x_id), *self.downsample[1].scale_shift(), relu=False)
        out = None
        if nhwc and folded is not None:
            out = pointwise.conv1x1(x, folded[0], self.bn1.scale_shift()[1], relu=True)
        if out is None:
            out = frozen_bn_act(self.conv1(x), *self.bn1.scale_shift(), relu=True)
        act1 = out
        out2 = None
        if self.conv2.stride == (1, 1):
            # 3x3 / stride 1 + frozen BN + ReLU: one Winograd/MFMA launch (csrc/wino.hip)
            out2 = conv3x3_bn_relu(out, self.conv2.weight, *self.bn2.scale_shift())
            if out2 is None:                       # wide layers: library forward / data gradient, own weight gradient
                y2 = conv3x3_own_wgrad(out, self.conv2.weight)
                if y2 is not None:
                    out2 = frozen_bn_act(y2, *self.bn2.scale_shift(), relu=True)
        elif nhwc and self.conv2.stride == (2, 2):
            # 3x3 / stride 2 + frozen BN + ReLU: the tap-list MFMA kernels (csrc/conv_tap.hip)
            out2 = strided.conv3x3_s2(out, self.conv2.weight, *self.bn2.scale_shift(), relu=True)
        out = out2 if out2 is not None else frozen_bn_act(self.conv2(out), *self.bn2.scale_shift(), relu=True)
        y3 = pointwise.conv1x1(out, self.conv3.weight) if nhwc else None
        if y3 is None:
            y3 = self.conv3(out)
        if self.pair_out and self.training:
            return frozen_bn_act(y3, *self.bn3.scale_shift(), residual=identity, relu=True, twice=True)
        y = frozen_bn_act(y3, *self.bn3.scale_shift(), residual=identity, relu=True)
        if _CAPTURE is not None:          # tests: the three activations, so that a float64 reference can open
            _CAPTURE.append((act1, out, y))   # its ReLUs where this run did (as datr_amd.bottleneck._CAPTURE)
        return y

    def fold_pairs(self):
        """(slot, weight, frozen scale) of the 1x1 convolutions whose batch norm is folded into the GEMM:
        slot 0 conv1, 1 the downsample convolution, 2 conv3 (the own bottleneck node)."""
        pairs = [(0, self.conv1.weight, self.bn1.scale_shift()[0])]
        if self.downsample is not None and self.downsample[0].stride in ((1, 1), (2, 2)):
            pairs.append((1, self.downsample[0].weight, self.downsample[1].scale_shift()[0]))
        pairs.append((2, self.conv3.weight, self.bn3.scale_shift()[0]))
        return pairs

    def own_node(self, x, gate_in: bool, gated_out: bool):
        """The whole block as one autograd node on the own kernels (datr_amd.bottleneck), or None when
        that node does not cover this block / input."""
        folded = getattr(self, "_folded", None)
        if folded is None or not isinstance(self.bn1, FrozenBatchNorm2d) or isinstance(x, tuple):
            return None
        stride = self.conv2.stride[0]
        if self.conv2.stride not in ((1, 1), (2, 2)) or not bottleneck.applicable(x, self.conv2.out_channels, stride):
            return None
        wds = shiftd = None
        if self.downsample is not None:
            if folded[1] is None or self.downsample[0].stride != self.conv2.stride:
                return None
            wds, shiftd = folded[1], self.downsample[1].scale_shift()[1]
        scale2, shift2 = self.bn2.scale_shift()
        return bottleneck.bottleneck(x, folded[0], self.conv2.weight, folded[2], wds, self.bn1.scale_shift()[1],
                                     scale2, shift2, self.bn3.scale_shift()[1], shiftd, stride, gate_in, gated_out)


class BottleneckStage(nn.Sequential):
    """A ResNet stage.  Blocks the own node covers are chained with the ReLU backward of a block's
    output applied by the NEXT block's data-gradient epilogue (datr_amd.bottleneck): a block's
    gradient then arrives gated (`gated_out`) and its consumer gates (`gate_in`)."""

    def forward(self, x):
        blocks = list(self)
        pair = isinstance(x, tuple)          # two handles on one tensor (fused.fan_out): conv1 / identity
        probe = x[0] if pair else x
        own = [False] * len(blocks)
        # which blocks take the own node is decided from the stage input (all blocks of a stage see
        # the same layout / dtype; the pixel count only shrinks at block 0)
        usable = (bottleneck.OWN_BOTTLENECK and isinstance(probe, torch.Tensor) and probe.is_cuda
                  and probe.dtype == torch.float32 and probe.dim() == 4 and not torch.is_autocast_enabled()
                  and probe.is_contiguous(memory_format=torch.channels_last) and not probe.is_contiguous())
        out_pixels = 0
        for i, blk in enumerate(blocks):
            if not (usable and isinstance(blk, Bottleneck) and getattr(blk, "_folded", None) is not None
                    and isinstance(blk.bn1, FrozenBatchNorm2d)):
                break
            stride = blk.conv2.stride[0]
            if i == 0:
                ok = blk.conv2.stride in ((1, 1), (2, 2)) and bottleneck.applicable(probe, blk.conv2.out_channels, stride) \
                    and (blk.downsample is None or (blk._folded[1] is not None and blk.downsample[0].stride == blk.conv2.stride))
                N, _, H, W = probe.shape
                out_pixels = N * ((H + stride - 1) // stride) * ((W + stride - 1) // stride)
            else:
                ok = (blk.downsample is None and blk.conv2.stride == (1, 1) and out_pixels >= bottleneck.MIN_PIXELS
                      and blk.conv2.out_channels % 64 == 0)
            own[i] = ok
            if not ok:
                break
        x_id = x
        if pair:
            x, x_id = x
            if own[0]:                       # the node takes ONE handle; the other one stays without a gradient
                x_id = x
        for i, blk in enumerate(blocks):
            y = None
            if own[i]:
                gate_in = i > 0 and own[i - 1] and torch.is_grad_enabled()
                gated_out = i + 1 < len(blocks) and own[i + 1] and torch.is_grad_enabled()
                y = blk.own_node(x, gate_in, gated_out)
                assert y is not None or not (gate_in or gated_out), "own bottleneck chain broken"
            if y is None:
                y = blk((x, x_id) if x is not x_id else x)
            x = x_id = y
            if isinstance(y, tuple):
                x, x_id = y
        return (x, x_id) if x is not x_id else x


class ResNet50Body(nn.Module):
    """conv1 / bn1 / relu / maxpool / layer1..4 with torchvision's child names and order."""

    def __init__(self, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or FrozenBatchNorm2d
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        for idx, (planes, blocks, stride) in enumerate(
                [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], start=1):
            layers = [Bottleneck(inplanes, planes, stride, norm_layer, downsample=True)]
            inplanes = planes * 4
            layers += [Bottleneck(inplanes, planes, 1, norm_layer, downsample=False)
                       for _ in range(blocks - 1)]
            for blk in layers[:-1]:                  # their successor has no downsample branch: see pair_out
                blk.pair_out = True
            setattr(self, f"layer{idx}", BottleneckStage(*layers))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def stem(self, x):
        if isinstance(self.bn1, FrozenBatchNorm2d):
            return frozen_bn_act(self.conv1(x), *self.bn1.scale_shift(), relu=True)
        return self.relu(self.bn1(self.conv1(x)))

    def forward(self, x):                      # plain classifier-less trunk
        x = self.maxpool(self.stem(x))
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return x


class _StageOutputs(nn.ModuleDict):
    """Runs the trunk's children in order and returns the requested stages
    (the role torchvision's IntermediateLayerGetter plays in backbone.py:96)."""

    def __init__(self, trunk: nn.Module, return_layers: Dict[str, str]):
        wanted = dict(return_layers)
        kept = OrderedDict()
        for name, child in trunk.named_children():
            kept[name] = child
            wanted.pop(name, None)
            if not wanted:
                break
        super().__init__(kept)
        self.return_layers = dict(return_layers)

    def _fold_bottlenecks(self, x):
        """Frozen-BN scales folded into the weights of every bottleneck's conv1 (and downsample
        convolution) for this forward pass -- ONE multi-tensor multiply for the whole
        trunk (datr_amd.pointwise.fold_frozen_bn); blocks read their share from `_folded`."""
        blocks = [m for m in self.modules() if isinstance(m, Bottleneck)]
        live = (pointwise.GEMM_1X1 and x.is_cuda and x.dtype == torch.float32
                and x.is_contiguous(memory_format=torch.channels_last)
                and all(isinstance(b.bn1, FrozenBatchNorm2d) for b in blocks))
        if not live:
            for b in blocks:
                b._folded = None
            return
        pairs, owner = [], []
        for b in blocks:
            for slot, w, sc in b.fold_pairs():
                pairs.append((w, sc))
                owner.append((b, slot))
        folded = pointwise.fold_frozen_bn(pairs)
        for b in blocks:
            b._folded = [None, None, None]
        for (b, k), f in zip(owner, folded):
            b._folded[k] = f

    def forward(self, x):
        self._fold_bottlenecks(x)
        try:
            return self._run(x)
        finally:                               # graph tensors must not outlive the pass as module
            for m in self.modules():           # attributes (copy.deepcopy of the model: EMA teacher)
                if isinstance(m, Bottleneck):
                    m._folded = None

    def _run(self, x):
        out = OrderedDict()
        items = list(self.items())
        i = 0
        while i < len(items):
            name, child = items[i]
            if (name == "conv1" and i + 2 < len(items)
                    and isinstance(items[i + 1][1], FrozenBatchNorm2d)
                    and isinstance(items[i + 2][1], nn.ReLU)):
                # stem: conv -> frozen BN -> ReLU in one launch (csrc/stem.hip) when the stem is
                # frozen, else the library convolution + one fused pass
                y = strided.stem_conv_bn_relu(x, child.weight, *items[i + 1][1].scale_shift()) \
                    if (child.bias is None and child.stride == (2, 2) and child.padding == (3, 3)) else None
                x = y if y is not None else frozen_bn_act(child(x), *items[i + 1][1].scale_shift(), relu=True)
                i += 3
                continue
            x = child(x)
            if name in self.return_layers:
                # a returned stage output has three kinds of consumers: the next stage's conv1, its
                # downsample branch, and the neck / discriminator -- one handle each, their gradients
                # meet in one pass (fused.fan_out) instead of pairwise adds over the feature map
                if i + 1 < len(items) and self.training and isinstance(x, torch.Tensor):
                    h = fan_out(x, 3)
                    out[self.return_layers[name]] = h[2]
                    x = (h[0], h[1]) if h[0] is not h[1] else x
                else:
                    out[self.return_layers[name]] = x
            i += 1
        return out


_EMPTY_MASKS = {}     # (batch, (H, W), device) -> all-False padding mask of an unpadded batch


class Backbone(nn.Module):
    """ResNet-50 trunk; conv1/bn1/layer1 frozen, layer2-4 trainable (backbone.py:79-81);
    returns {name: NestedTensor(feature, nearest-resized padding mask)}."""

    def __init__(self, name: str, train_backbone: bool, dilation: bool,
                 return_interm_indices: List[int], batch_norm=FrozenBatchNorm2d):
        super().__init__()
        if name != "resnet50":
            raise NotImplementedError(f"only resnet50 is on the hot path, got {name}")
        if dilation:
            raise NotImplementedError("dilation=False in every DA config")
        assert return_interm_indices in [[0, 1, 2, 3], [1, 2, 3], [3]]
        trunk = ResNet50Body(norm_layer=batch_norm)
        for pname, p in trunk.named_parameters():
            if not train_backbone or not any(k in pname for k in ("layer2", "layer3", "layer4")):
                p.requires_grad_(False)
        n = len(return_interm_indices)
        return_layers = {f"layer{5 - n + i}": str(idx) for i, idx in enumerate(return_interm_indices)}
        self.body = _StageOutputs(trunk, return_layers)
        self.num_channels = [256, 512, 1024, 2048][4 - n:]

    def _nhwc_trunk(self) -> bool:
        """True when the trunk was moved to torch.channels_last (its first convolution's weight says so)."""
        for mod in self.body.modules():
            if isinstance(mod, nn.Conv2d):
                w = mod.weight
                return w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous()
        return False

    def forward(self, tensor_list: NestedTensor):
        x = tensor_list.tensors
        if (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
                and not x.is_contiguous(memory_format=torch.channels_last) and self._nhwc_trunk()):
            # an NCHW batch (the reference's collate function builds one, util/misc.py:387-409) into an NHWC trunk:
            # one pass over the images here instead of the whole backbone on the library's NCHW path
            x = x.contiguous(memory_format=torch.channels_last)
        feats = self.body(x)
        out: Dict[str, NestedTensor] = {}
        m = tensor_list.mask
        assert m is not None
        for name, x in feats.items():
            if tensor_list.padded is False and m.is_cuda:
                # no padded pixel (known on the host): the resized mask is all False whatever the size
                key = (int(m.shape[0]), tuple(x.shape[-2:]), str(m.device))
                mask = _EMPTY_MASKS.get(key)
                if mask is None:
                    if len(_EMPTY_MASKS) >= 32:
                        _EMPTY_MASKS.clear()
                    mask = _EMPTY_MASKS[key] = torch.zeros((key[0],) + key[1], dtype=torch.bool, device=m.device)
            else:
                mask = F.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]
            out[name] = NestedTensor(x, mask, tensor_list.padded)
        return out


class PositionEmbeddingSineHW(nn.Module):
    """Sine/cosine embedding of the cumulative (unpadded) row / column index, normalised to
    [0, 2*pi], with separate temperatures for y and x; output [N, 2*num_pos_feats, H, W] with
    the y half first (position_encoding.py:79-108)."""

    def __init__(self, num_pos_feats=64, temperatureH=10000, temperatureW=10000, normalize=False,
                 scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperatureH, self.temperatureW = temperatureH, temperatureW
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def forward(self, tensor_list: NestedTensor):
        x, mask = tensor_list.tensors, tensor_list.mask
        assert mask is not None
        if getattr(tensor_list, "padded", None) is False:
            # no padded pixel (known on the host): the embedding depends on the shape only --
            # computed once per (N, H, W) instead of ~15 launches per level and pass
            key = (tuple(mask.shape), str(x.device), self.num_pos_feats, self.temperatureH,
                   self.temperatureW, self.normalize, self.scale)
            hit = self._cache.get(key)
            if hit is None:
                if len(self._cache) >= 16:
                    self._cache.clear()
                hit = self._cache[key] = self._embed(x, mask).detach()
            return hit
        return self._embed(x, mask)

    def _embed(self, x, mask):
        valid = ~mask
        y_embed = valid.cumsum(1, dtype=torch.float32)
        x_embed = valid.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        k = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        expo = 2 * torch.div(k, 2, rounding_mode="floor") / self.num_pos_feats
        px = x_embed[:, :, :, None] / (self.temperatureW ** expo)
        py = y_embed[:, :, :, None] / (self.temperatureH ** expo)
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


class Joiner(nn.Sequential):
    """[0] = Backbone, [1] = position embedding (state_dict prefix `backbone.0.body...`)."""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list: NestedTensor):
        feats = self[0](tensor_list)
        out: List[NestedTensor] = []
        pos = []
        for _, x in feats.items():
            out.append(x)
            pos.append(self[1](x).to(x.tensors.dtype))
        return out, pos


def build_position_encoding(args):
    if args.position_embedding not in ("v2", "sine"):
        raise ValueError(f"not supported {args.position_embedding}")
    return PositionEmbeddingSineHW(args.hidden_dim // 2, temperatureH=args.pe_temperatureH,
                                   temperatureW=args.pe_temperatureW, normalize=True)


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    if not args.lr_backbone > 0:
        raise ValueError("Please set lr_backbone > 0")
    if args.backbone not in ("resnet50",):
        raise NotImplementedError(f"Unknown backbone {args.backbone}")
    backbone = Backbone(args.backbone, True, args.dilation, args.return_interm_indices,
                        batch_norm=FrozenBatchNorm2d)
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    return model
