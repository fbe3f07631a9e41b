"""The product's ResNet-50 trunk (datr_amd/backbone.py: frozen batch norm folded into affine passes / GEMM
epilogues, whole bottlenecks as single nodes on the device) against the oracle's plain-nn restatement of
torchvision's ResNet-50 v1.5 (oracle/resnet_ref.py -- the stand-in the golden generators hand to the
reference's `Backbone`, /root/reference/models/dino/backbone.py:109-128).  This file: names, shapes, trainable
split and the host-side arithmetic; tests/test_backbone_gpu.py: the device kernels."""
import torch

from helpers import ROOT  # noqa: F401  (puts tests/golden on sys.path)
import synth
from datr_amd import backbone as B
from oracle import resnet_ref


class RefFrozenBN(torch.nn.Module):
    """backbone.py:36-72 restated: y = x * w * rsqrt(var + 1e-5) + (b - mean * w * rsqrt(var + 1e-5))."""

    def __init__(self, n):
        super().__init__()
        for name, v in (("weight", torch.ones(n)), ("bias", torch.zeros(n)),
                        ("running_mean", torch.zeros(n)), ("running_var", torch.ones(n))):
            self.register_buffer(name, v)

    def forward(self, x):
        w, b = self.weight.reshape(1, -1, 1, 1), self.bias.reshape(1, -1, 1, 1)
        rv, rm = self.running_var.reshape(1, -1, 1, 1), self.running_mean.reshape(1, -1, 1, 1)
        scale = w * (rv + 1e-5).rsqrt()
        return x * scale + (b - rm * scale)


def ref_trunk(dtype=torch.float32):
    ref = resnet_ref.resnet50(replace_stride_with_dilation=[False, False, False], pretrained=True,
                              norm_layer=RefFrozenBN)
    synth.synth_init_(ref)          # keyed by parameter name: the product's trunk gets the same tensors
    return ref.to(dtype)


def ref_features(ref, x):
    x = ref.maxpool(ref.relu(ref.bn1(ref.conv1(x))))
    c2 = ref.layer1(x)
    c3 = ref.layer2(c2)
    c4 = ref.layer3(c3)
    return c3, c4, ref.layer4(c4)


def test_state_dict_names_shapes_and_module_order_match_torchvision_layout():
    ref, mine = ref_trunk(), B.ResNet50Body()
    rs = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.startswith("fc.")}
    ms = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert list(rs) == list(ms) and rs == ms
    assert [n for n, _ in ref.named_children()][:8] == [n for n, _ in mine.named_children()]
    # v1.5: the stride sits on the 3x3 convolution of the first block of layers 2-4
    for l in (2, 3, 4):
        blk = getattr(mine, f"layer{l}")[0]
        assert blk.conv1.stride == (1, 1) and blk.conv2.stride == (2, 2) and blk.downsample[0].stride == (2, 2)


def test_trainable_split_is_the_references():
    """backbone.py:79-81: conv1 / layer1 frozen, layers 2-4 trained (batch-norm tensors are buffers)."""
    bb = B.Backbone("resnet50", True, False, [1, 2, 3])
    for n, p in bb.named_parameters():
        assert p.requires_grad == any(f"layer{i}" in n for i in (2, 3, 4)), n
    assert bb.num_channels == [512, 1024, 2048]


def test_host_forward_and_backward_match_the_plain_restatement():
    torch.manual_seed(0)
    ref, mine = ref_trunk(), B.ResNet50Body()
    synth.synth_init_(mine)
    x = torch.randn(2, 3, 96, 128)
    feats_ref = ref_features(ref, x)
    mine.train()
    y = mine.maxpool(mine.stem(x))
    feats = []
    for i in range(1, 5):
        y = getattr(mine, f"layer{i}")(y)
        y = y[0] if isinstance(y, tuple) else y      # a stage may hand over a (conv1, identity) handle pair
        feats.append(y)
    for a, b in zip(feats[1:], feats_ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    g = [torch.randn_like(f) for f in feats_ref]
    names = [n for n, p in ref.named_parameters() if "layer" in n and not n.startswith("layer1")]
    gr = torch.autograd.grad(sum((f * w).sum() for f, w in zip(feats_ref, g)),
                             [dict(ref.named_parameters())[n] for n in names])
    gm = torch.autograd.grad(sum((f * w).sum() for f, w in zip(feats[1:], g)),
                             [dict(mine.named_parameters())[n] for n in names])
    for n, a, b in zip(names, gm, gr):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()), msg=n)
