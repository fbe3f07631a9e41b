"""Stride-2 convolutions on the own MFMA kernels (csrc/conv_tap.hip) against float64 autograd.

The layers: conv2 (3x3) and the downsample convolution (1x1) of layer2.0 / layer3.0 / layer4.0 of the
ResNet-50 trunk (/root/reference/models/dino/backbone.py:109-128) with their frozen batch norm and
ReLU, and input_proj[3] (/root/reference/models/dino/dino.py:120-124) with its bias.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, scale, shift, relu):
    y = F.conv2d(x, w, stride=2, padding=w.shape[-1] // 2)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    return y.relu() if relu else y


CASES = [
    # N, Cin, Cout, H, W, k, bn, relu, bias            what
    (2, 128, 128, 50, 67, 3, True, True, False),      # layer2.0.conv2 (odd width, partial tiles)
    (2, 256, 256, 26, 34, 3, True, True, False),      # layer3.0.conv2
    (4, 512, 512, 25, 42, 3, True, True, False),      # layer4.0.conv2 at its step size (split K)
    (4, 2048, 256, 25, 42, 3, False, False, True),    # input_proj[3] at its step size (split K x 16)
    (2, 256, 512, 40, 56, 1, True, False, False),     # layer2.0.downsample
    (2, 1024, 2048, 13, 21, 1, True, False, False),   # layer4.0.downsample
    (2, 128, 128, 256, 512, 3, True, True, False),    # layer2.0.conv2 at the C2F geometry (1024 x 2048 frames)
    (1, 128, 128, 1, 1, 3, False, False, False),      # a single pixel
    (1, 128, 128, 8, 17, 3, False, False, True),
]


@pytest.mark.parametrize("N,ci,co,H,W,k,bn,relu,bias", CASES)
def test_conv_s2_matches_float64_autograd(N, ci, co, H, W, k, bn, relu, bias):
    from datr_amd.strided import conv1x1_s2, conv3x3_s2
    g = torch.Generator().manual_seed(H * 131 + W)
    dev = torch.device("cuda:0")
    x = torch.randn(N, ci, H, W, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
    scale = (torch.rand(co, generator=g) + 0.5) if bn else None
    shift = torch.randn(co, generator=g) * 0.1 if (bn or bias) else None
    gy = torch.randn(N, co, (H + 1) // 2, (W + 1) // 2, generator=g)

    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    sd = None if shift is None else shift.double().requires_grad_(True)
    yd = _ref(xd, wd, None if scale is None else scale.double(), sd, relu)
    yd.backward(gy.double())

    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = w.to(dev).requires_grad_(True)
    sg = None if shift is None else shift.to(dev).requires_grad_(bias)
    if k == 3:
        y = conv3x3_s2(xg, wg, None if scale is None else scale.to(dev), sg, relu)
    else:
        # the downsample path: frozen scale folded into the weight, shift as the GEMM's bias
        y = conv1x1_s2(xg, wg * scale.to(dev).view(-1, 1, 1, 1), sg, relu)
    assert y is not None and y.shape == yd.shape
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(dev))
    torch.cuda.synchronize()

    def close(a, b, what):
        b = b.float()
        err = (a.cpu() - b).abs().max().item()
        ref = b.abs().max().item()
        assert err <= 2e-5 * max(ref, 1.0) + 1e-5 * ref, f"{what}: max err {err:.3e} (scale {ref:.3e})"

    close(y.detach(), yd.detach(), "forward")
    close(xg.grad, xd.grad, "data gradient")
    close(wg.grad, wd.grad, "weight gradient")
    if bias:
        close(sg.grad, sd.grad, "bias gradient")


def test_conv_s2_is_bitwise_reproducible():
    from datr_amd.strided import conv1x1_s2, conv3x3_s2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 512, 25, 42, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(512, 512, 3, 3, generator=g) / 68).to(dev)
    gy = torch.randn(4, 512, 13, 21, generator=g).to(dev)
    outs = []
    for _ in range(2):
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = conv3x3_s2(xg, wg)
        y.backward(gy)
        outs.append((y.detach().clone(), xg.grad.clone(), wg.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 37, 53), (2, 200, 333), (1, 1024, 2048)])
def test_stem_matches_float64(N, H, W):
    """conv1 + bn1 + relu of the frozen trunk head (backbone.py:62-72,79-81) in one launch."""
    from datr_amd.strided import stem_conv_bn_relu
    g = torch.Generator().manual_seed(H + W)
    dev = torch.device("cuda:0")
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.1
    ref = (F.conv2d(x.double(), w.double(), stride=2, padding=3) * scale.double().view(1, -1, 1, 1)
           + shift.double().view(1, -1, 1, 1)).relu()
    y = stem_conv_bn_relu(x.to(dev).contiguous(memory_format=torch.channels_last), w.to(dev), scale.to(dev), shift.to(dev))
    assert y is not None and y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    err = (y.cpu() - ref.float()).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err
