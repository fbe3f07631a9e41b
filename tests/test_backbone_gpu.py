"""GPU: the whole ResNet-50 trunk on the device path (NHWC, frozen batch norm folded into GEMM epilogues,
Winograd / tap-list MFMA convolutions, whole bottlenecks as single autograd nodes) against the ORACLE's
plain-nn ResNet-50 v1.5 (oracle/resnet_ref.py, float64 on the host cores) -- the class the golden generators
hand to the reference's `Backbone` (/root/reference/models/dino/backbone.py:109-128).  A wiring mistake in
the product's trunk (residual order, stride placement, downsample branch) cannot hide here: the two sides
share no code."""
import pytest
import torch

from helpers import ROOT  # noqa: F401
from test_backbone_cpu import ref_features, ref_trunk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nhwc", [False, True])
def test_trunk_features_and_weight_gradients_match_plain_resnet50(nhwc, monkeypatch):
    from datr_amd import backbone as B, bottleneck
    from datr_amd.nested import NestedTensor
    monkeypatch.setattr(bottleneck, "MIN_PIXELS", 1)          # the own nodes at this small image too
    dev = torch.device("cuda:0")
    ref = ref_trunk(torch.float64)
    bb = B.Backbone("resnet50", True, False, [1, 2, 3])
    sd = {k: v.float() for k, v in ref.state_dict().items() if not k.startswith("fc.")}
    bb.body.load_state_dict(sd, strict=True)
    bb = bb.to(dev)
    torch.manual_seed(1)
    x = torch.randn(2, 3, 160, 224)
    xd = x.to(dev)
    if nhwc:
        bb.to(memory_format=torch.channels_last)
        xd = xd.contiguous(memory_format=torch.channels_last)
    bb.train()
    out = bb(NestedTensor(xd, torch.zeros(2, 160, 224, dtype=torch.bool, device=dev)))
    feats = [out[k].tensors for k in ("1", "2", "3")]
    feats_ref = ref_features(ref, x.double())
    for a, b in zip(feats, feats_ref):
        assert a.shape == b.shape
        tol = 1e-3 * float(b.abs().max())
        torch.testing.assert_close(a.detach().double().cpu(), b.detach(), rtol=1e-3, atol=tol)
    # weight gradients of the trainable stages; a pre-activation within fp32 rounding of zero gates
    # differently in the two precisions, so compare tensor-wise in relative L2
    torch.manual_seed(2)
    g = [torch.randn_like(f) for f in feats_ref]
    names = [n for n, p in ref.named_parameters() if n.startswith(("layer2", "layer3", "layer4"))]
    gr = torch.autograd.grad(sum((f * w).sum() for f, w in zip(feats_ref, g)),
                             [dict(ref.named_parameters())[n] for n in names])
    mine_p = dict(bb.body.named_parameters())
    gm = torch.autograd.grad(sum((f * w.to(dev).float()).sum() for f, w in zip(feats, g)),
                             [mine_p[n] for n in names])
    worst = 0.0
    for n, a, b in zip(names, gm, gr):
        err = float((a.double().cpu() - b).norm() / b.norm())
        worst = max(worst, err)
        assert err < 2e-2, (n, err)
    assert all(p.grad is None for n, p in mine_p.items() if n.startswith(("conv1", "layer1")))
