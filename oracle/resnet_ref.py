"""oracle/resnet_ref.py -- TEST INFRASTRUCTURE: a plain-`nn` restatement of torchvision's ResNet-50.

The reference takes its backbone arithmetic from an un-vendored third-party package:
`getattr(torchvision.models, name)(replace_stride_with_dilation=[False, False, dilation],
pretrained=..., norm_layer=FrozenBatchNorm2d)` (/root/reference/models/dino/backbone.py:118-120,
requirements.txt:5 `torchvision>=0.6.0`, README env 0.15.2).  torchvision is absent from this image, so
the golden generators (tests/golden/make_golden*.py) need a stand-in for `torchvision.models.resnet50`.
This file IS that stand-in, written from torchvision's published architecture (ResNet "v1.5": the
stride of a bottleneck sits on its 3x3 convolution) with nothing but `nn.Conv2d`, the caller's
`norm_layer`, `nn.ReLU`, `nn.MaxPool2d`, `nn.AdaptiveAvgPool2d` and `nn.Linear`:

  * child names and registration order of torchvision 0.15's `ResNet`: conv1, bn1, relu, maxpool,
    layer1..layer4, avgpool, fc (the reference's IntermediateLayerGetter walks `named_children()` in
    that order and stops after layer4); block children conv1, bn1, conv2, bn2, conv3, bn3, relu,
    downsample -- hence the state_dict names of SURVEY.md A.2 (`layerK.J.{convN,bnN,downsample.{0,1}}`);
  * construction order as torchvision's `_make_layer` (a stage's downsample branch is built before its
    first block) and its initialisation pass (`kaiming_normal_(fan_out, relu)` over every convolution in
    `modules()` order, constants for nn.BatchNorm2d / nn.GroupNorm), so that the RNG stream a seeded
    `build_dino` sees behind the backbone matches a real torchvision install.

It deliberately imports NOTHING from `datr_amd` (tests/test_native_abi.py checks that for the whole of
oracle/ and tests/golden/): the fixtures compare the product with this file, never with itself.
Parity of this restatement with torchvision itself stays unpinned (no torchvision here to run); what it
pins is that the product's fused / NHWC / MFMA backbone computes what a plain conv-bn-relu ResNet-50 does.
"""
from __future__ import annotations

import torch
from torch import nn


def _conv3x3(cin: int, cout: int, stride: int = 1, dilation: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                     bias=False)


def _conv1x1(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


class RefBottleneck(nn.Module):
    """1x1 reduce -> 3x3 (carries the stride: v1.5) -> 1x1 expand (x4), identity or projected shortcut."""

    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = _conv1x1(inplanes, planes)
        self.bn1 = norm_layer(planes)
        self.conv2 = _conv3x3(planes, planes, stride, dilation)
        self.bn2 = norm_layer(planes)
        self.conv3 = _conv1x1(planes, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y = y + shortcut
        return self.relu(y)


class RefResNet50(nn.Module):
    """ResNet-50: stem 7x7/2 + maxpool 3x3/2, stages of (3, 4, 6, 3) bottlenecks with 64/128/256/512
    planes, strides (1, 2, 2, 2); global average pool + 1000-way classifier (built for name / RNG
    fidelity; the detector never runs them)."""

    def __init__(self, norm_layer=None, replace_stride_with_dilation=None, num_classes=1000):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        dilate = replace_stride_with_dilation or [False, False, False]
        assert len(dilate) == 3
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(64, 3)
        self.layer2 = self._stage(128, 4, stride=2, dilate=dilate[0])
        self.layer3 = self._stage(256, 6, stride=2, dilate=dilate[1])
        self.layer4 = self._stage(512, 3, stride=2, dilate=dilate[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * RefBottleneck.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * RefBottleneck.expansion:
            downsample = nn.Sequential(_conv1x1(self.inplanes, planes * RefBottleneck.expansion, stride),
                                       norm_layer(planes * RefBottleneck.expansion))
        layers = [RefBottleneck(self.inplanes, planes, stride, downsample, previous_dilation, norm_layer)]
        self.inplanes = planes * RefBottleneck.expansion
        for _ in range(1, blocks):
            layers.append(RefBottleneck(self.inplanes, planes, dilation=self.dilation, norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50(replace_stride_with_dilation=None, pretrained=False, norm_layer=None, weights=None, **kw):
    """Signature of the call at backbone.py:118-120.  The reference passes `pretrained=is_main_process()`
    (True on rank 0), which in torchvision downloads the ImageNet weights; there is no network here, so the
    flag is accepted and ignored -- every fixture overwrites all weights with tests/golden/synth.py's tensors."""
    del pretrained, weights, kw
    return RefResNet50(norm_layer=norm_layer, replace_stride_with_dilation=replace_stride_with_dilation)
