"""A ResNet stage on the own bottleneck node (datr_amd/bottleneck.py: the whole block as one autograd
node, ReLU backward of a block's output applied by the next block's data-gradient epilogue) against the
reference's op sequence in float64 under autograd: torchvision's Bottleneck with FrozenBatchNorm2d
(/root/reference/models/dino/backbone.py:36-72,109-128)."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference_stage(blocks, x, masks=None):
    """The blocks' arithmetic op by op in the tensors' dtype (float64 here).  masks: per block the three
    activations of the float32 run -- the reference then opens its ReLUs where THAT run did (a
    pre-activation within rounding of zero gates differently in the two precisions, which changes single
    gradient entries by O(1): not an error of either side)."""
    def bn(m, t):
        scale = m.weight * (m.running_var + 1e-5).rsqrt()
        return t * scale.view(1, -1, 1, 1) + (m.bias - m.running_mean * scale).view(1, -1, 1, 1)

    def relu(t, k, j):
        return torch.relu(t) if masks is None else t * (masks[k][j] > 0).to(t.dtype)
    for k, b in enumerate(blocks):
        idn = x if b.downsample is None else bn(b.downsample[1], F.conv2d(x, b.downsample[0].weight, stride=b.downsample[0].stride))
        o = relu(bn(b.bn1, F.conv2d(x, b.conv1.weight)), k, 0)
        o = relu(bn(b.bn2, F.conv2d(o, b.conv2.weight, stride=b.conv2.stride, padding=1)), k, 1)
        x = relu(bn(b.bn3, F.conv2d(o, b.conv3.weight)) + idn, k, 2)
    return x


def _make_stage(inplanes, planes, nblocks, stride, dev, seed):
    from datr_amd.backbone import Bottleneck, BottleneckStage, FrozenBatchNorm2d
    torch.manual_seed(seed)
    layers = [Bottleneck(inplanes, planes, stride, FrozenBatchNorm2d, downsample=True)]
    layers += [Bottleneck(planes * 4, planes, 1, FrozenBatchNorm2d, downsample=False) for _ in range(nblocks - 1)]
    stage = BottleneckStage(*layers)
    for m in stage.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0)
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return stage.to(dev)


def _fold(stage, x):
    """What _StageOutputs._fold_bottlenecks does for the trunk."""
    from datr_amd import pointwise
    pairs, owner = [], []
    for b in stage:
        for slot, w, sc in b.fold_pairs():
            pairs.append((w, sc)); owner.append((b, slot))
    folded = pointwise.fold_frozen_bn(pairs)
    for b in stage:
        b._folded = [None, None, None]
    for (b, k), f in zip(owner, folded):
        b._folded[k] = f


@pytest.mark.parametrize("inplanes,planes,nblocks,stride,shape,input_grad", [
    (256, 128, 3, 2, (2, 67, 90), True),        # layer2-like: stride-2 first block, odd sizes
    (64, 64, 3, 1, (2, 72, 70), False),         # layer1-like: stride-1 downsample, input without gradient
    (512, 256, 2, 2, (3, 66, 80), True),
])
def test_stage_matches_float64_autograd(inplanes, planes, nblocks, stride, shape, input_grad, monkeypatch):
    from datr_amd import bottleneck
    monkeypatch.setattr(bottleneck, "MIN_PIXELS", 1)
    dev = torch.device("cuda:0")
    stage = _make_stage(inplanes, planes, nblocks, stride, dev, seed=inplanes + planes)
    stage.train()
    N, H, W = shape
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn(N, inplanes, H, W, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(input_grad)
    _fold(stage, x)
    calls, acts = [], []
    orig = bottleneck._BottleneckFn.apply
    monkeypatch.setattr(bottleneck._BottleneckFn, "apply", lambda *a: (calls.append((a[-2], a[-1])), orig(*a))[1])
    monkeypatch.setattr(bottleneck, "_CAPTURE", acts)
    y = stage(x)
    assert calls == [(i > 0, i + 1 < nblocks) for i in range(nblocks)]        # every block on the node, chained
    go = torch.randn(y.shape, generator=g).to(dev)
    y.backward(go.contiguous(memory_format=torch.channels_last))
    for b in stage:
        b._folded = None

    ref_stage = copy.deepcopy(stage).double()
    for p in ref_stage.parameters():
        p.grad = None
    xr = x.detach().double().requires_grad_(input_grad)
    yr = _reference_stage(list(ref_stage), xr, masks=acts)
    yr.backward(go.double())

    errs = {}

    def close(name, a, b, tol):
        scale = max(float(b.abs().max()), 1e-30)
        err = float((a.detach().double() - b).abs().max()) / scale
        if not err < tol:
            errs[name] = err
    close("y", y, yr.detach(), 5e-6)
    if input_grad:
        close("x.grad", x.grad, xr.grad, 1e-5)
    for (n, p), (_, pr) in zip(stage.named_parameters(), ref_stage.named_parameters()):
        assert p.grad is not None, n
        close(n, p.grad, pr.grad, 1e-5)
    assert not errs, errs


def test_eval_mode_and_small_maps_take_a_consistent_path(monkeypatch):
    """No-grad forward: the node runs without gate flags; below MIN_PIXELS the per-op path runs; both
    agree with the float64 reference."""
    from datr_amd import bottleneck
    dev = torch.device("cuda:0")
    stage = _make_stage(256, 128, 2, 2, dev, seed=3).eval()
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(2, 256, 40, 44, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    ref = _reference_stage(list(copy.deepcopy(stage).double()), x.double())
    outs = []
    for min_pixels in (1, 1 << 30):
        monkeypatch.setattr(bottleneck, "MIN_PIXELS", min_pixels)
        with torch.no_grad():
            _fold(stage, x)
            outs.append(stage(x))
            for b in stage:
                b._folded = None
    for o in outs:
        assert float((o.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
