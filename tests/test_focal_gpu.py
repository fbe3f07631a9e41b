"""GPU: the fused focal-loss HIP kernels (through the C ABI) against the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G,R,C,alpha,gamma", [
    (7, 1800, 9, 0.25, 2.0),      # final + 5 aux + interm, B*Q = 2*900 (BASELINE shape)
    (6, 400, 9, 0.25, 2.0),       # de-noising family, B*pad = 2*200
    (1, 77, 2, 0.25, 2.0),        # Sim10k-style 2 classes, ragged
    (3, 1000, 9, -1.0, 2.0),      # alpha disabled
    (2, 513, 9, 0.5, 1.5),        # generic gamma
    (1, 20000, 91, 0.25, 2.0),    # many rows, COCO-sized class count
])
def test_focal_forward_backward_match_oracle(G, R, C, alpha, gamma):
    from datr_amd.focal import sigmoid_focal_loss_sums
    from oracle import focal_oracle as FO
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(G * 1000 + R)
    logits = (torch.randn(G, R, C, generator=g) * 4).requires_grad_(True)
    target = torch.randint(0, C + 1, (G, R), generator=g)
    w = torch.rand(G, generator=g) + 0.5
    lg = logits.detach().to(dev).requires_grad_(True)
    sums = sigmoid_focal_loss_sums(lg, target.to(dev), alpha, gamma)
    (sums * w.to(dev)).sum().backward()
    ref = FO.focal_sums_torch(logits, target, alpha, gamma)
    (ref * w).sum().backward()
    torch.testing.assert_close(sums.cpu().double(), FO.focal_sums_c(logits, target, alpha, gamma),
                               rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(sums.cpu(), ref.detach(), rtol=2e-5, atol=1e-5)
    torch.testing.assert_close(lg.grad.cpu(), logits.grad, rtol=1e-4, atol=1e-6)
    # deterministic: fixed-order fold, no float atomics
    assert torch.equal(sums, sigmoid_focal_loss_sums(lg.detach(), target.to(dev), alpha, gamma))


def test_focal_extreme_logits_and_empty():
    from datr_amd.focal import sigmoid_focal_loss_sums
    dev = torch.device("cuda:0")
    logits = torch.tensor([[[80.0, -80.0, 0.0], [-30.0, 30.0, 5.0]]], device=dev, requires_grad=True)
    target = torch.tensor([[0, 3]], device=dev)
    s = sigmoid_focal_loss_sums(logits, target, 0.25, 2.0)
    s.sum().backward()
    assert torch.isfinite(s).all() and torch.isfinite(logits.grad).all()
    empty = sigmoid_focal_loss_sums(torch.zeros(4, 0, 9, device=dev), torch.zeros(4, 0, dtype=torch.long, device=dev))
    assert empty.shape == (4,) and torch.count_nonzero(empty) == 0
    with pytest.raises(RuntimeError, match="CPU"):
        sigmoid_focal_loss_sums(torch.zeros(1, 2, 3), torch.zeros(1, 2, dtype=torch.long))


def _torch_box_losses(src, tgt, group, G):
    """The op sequence of SetCriterion.loss_boxes (dino.py:553-577) per prediction set."""
    import torch.nn.functional as F
    from datr_amd import boxes as box_ops
    l1 = F.l1_loss(src, tgt, reduction="none")
    giou = box_ops.generalized_box_iou_pairs(box_ops.box_cxcywh_to_xyxy(src), box_ops.box_cxcywh_to_xyxy(tgt))
    per_g = lambda v: torch.zeros(G, device=src.device).index_add_(0, group, v)
    return torch.stack([per_g(l1.sum(-1)), per_g(1 - giou), per_g(l1[..., :2].sum(-1)),
                        per_g(l1[..., 2:].sum(-1))])


@pytest.mark.parametrize("P,G", [(140, 7), (1200, 6), (3, 1), (3072, 13)])
def test_fused_box_losses_match_torch_ops(P, G):
    from datr_amd.focal import box_loss_sums
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(P + G)
    src = torch.cat([torch.rand(P, 2, generator=gen), torch.rand(P, 2, generator=gen) * 0.4 + 0.01], 1)
    tgt = torch.cat([torch.rand(P, 2, generator=gen), torch.rand(P, 2, generator=gen) * 0.4 + 0.01], 1)
    if P >= 3:
        src[0] = tgt[0]                                   # identical boxes: every max / min ties
        src[1] = torch.tensor([0.1, 0.1, 0.05, 0.05])     # disjoint boxes: empty intersection
        tgt[1] = torch.tensor([0.8, 0.8, 0.10, 0.10])
        src[2, 0] = tgt[2, 0]                             # one coordinate equal: |x|' at 0
    group = torch.randint(0, G, (P,), generator=gen)
    src, tgt, group = src.to(dev), tgt.to(dev), group.to(dev)
    a = src.clone().requires_grad_(True)
    b = src.clone().requires_grad_(True)
    fused = box_loss_sums(a, tgt, group, G)
    ref = _torch_box_losses(b, tgt, group, G)
    torch.testing.assert_close(fused, ref, rtol=2e-5, atol=2e-5)
    w = torch.rand(2, G, generator=gen).to(dev) + 0.5
    (fused[:2] * w).sum().backward()
    (ref[:2] * w).sum().backward()
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-5)
    again = box_loss_sums(src, tgt, group, G)
    assert torch.equal(again, fused.detach())             # deterministic
