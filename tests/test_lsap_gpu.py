"""GPU: the device Hungarian solver (csrc/lsap.hip) against the reference's own solver,
scipy.optimize.linear_sum_assignment (/root/reference/models/dino/matcher.py:20,94), on the
[num_queries x T] problems the matcher builds -- indices must be IDENTICAL, including on
matrices full of ties (integer costs, duplicated boxes), which pin SciPy's scan order and tie
rule."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu


def run(C, sizes):
    from datr_amd.matcher import solve_lsap_device
    q, t, status = solve_lsap_device(C.cuda(), sizes)
    return q.cpu(), t.cpu(), status.cpu()


def check(C, sizes):
    q, t, status = run(C, sizes)
    assert int(status.abs().sum()) == 0
    G, B = C.shape[:2]
    for g in range(G):
        off = 0
        for b, n in enumerate(sizes):
            rows, cols = linear_sum_assignment(C[g, b, :, off:off + n].numpy())
            assert q[g, off:off + n].tolist() == rows.tolist(), (g, b, n)
            assert t[g, off:off + n].tolist() == cols.tolist(), (g, b, n)
            off += n


@pytest.mark.parametrize("nq,sizes,G", [
    (900, [10, 10], 7),            # the bench step: 7 prediction sets x 2 images
    (900, [1, 0, 33, 7], 2),       # empty image, single box, LDS-staging boundary (33 x 900 x 4 > 96 KB)
    (900, [100, 20], 1),           # cost rows re-read from global memory
    (300, [299, 5], 1),            # almost square (T == nq is SciPy's untransposed case: host path)
    (1024, [64], 3),
    (17, [3, 16], 2),
    (900, [300, 150, 1], 1),       # many boxes: long augmenting paths, rows read from global memory
])
def test_random_costs(nq, sizes, G):
    g = torch.Generator().manual_seed(nq + sum(sizes))
    C = torch.randn(G, len(sizes), nq, sum(sizes), generator=g) * 3.0
    check(C, sizes)


@pytest.mark.parametrize("nq,sizes,levels", [(900, [12, 9], 3), (200, [40, 40], 2), (64, [63], 4),
                                             (900, [20], 1)])
def test_tie_heavy_integer_costs(nq, sizes, levels):
    """Few distinct integer costs => massive ties; a constant matrix (levels == 1) must give
    SciPy's identity-like answer."""
    g = torch.Generator().manual_seed(levels)
    C = torch.randint(0, levels, (3, len(sizes), nq, sum(sizes)), generator=g).float()
    check(C, sizes)


def test_duplicated_boxes_and_matcher_costs():
    """Real matcher costs with duplicated ground-truth boxes (identical columns)."""
    from datr_amd.matcher import HungarianMatcher
    g = torch.Generator().manual_seed(5)
    m = HungarianMatcher(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
    B, nq, ncls = 2, 900, 9
    out = {"pred_logits": torch.randn(B, nq, ncls, generator=g),
           "pred_boxes": torch.cat([torch.rand(B, nq, 2, generator=g) * 0.6 + 0.2,
                                    torch.rand(B, nq, 2, generator=g) * 0.3 + 0.02], -1)}
    targets = []
    for b in range(B):
        box = torch.cat([torch.rand(6, 2, generator=g) * 0.6 + 0.2, torch.rand(6, 2, generator=g) * 0.2 + 0.05], -1)
        lab = torch.randint(1, 9, (6,), generator=g)
        targets.append({"boxes": torch.cat([box, box[:3]]), "labels": torch.cat([lab, lab[:3]])})
    host = m(out, targets)                                         # CPU tensors -> SciPy path
    dev = torch.device("cuda:0")
    dout = {k: v.to(dev) for k, v in out.items()}
    dtargets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    got = m(dout, dtargets)
    for (hi, hj), (gi, gj) in zip(host, got):
        assert gi.is_cuda and gi.cpu().tolist() == hi.tolist() and gj.cpu().tolist() == hj.tolist()
    many = m.forward_many([dout, dout], dtargets)
    for per in many:
        for (hi, hj), (gi, gj) in zip(host, per):
            assert gi.cpu().tolist() == hi.tolist() and gj.cpu().tolist() == hj.tolist()
    assert float(m.poison) == 0.0


def test_invalid_costs_are_flagged():
    C = torch.randn(1, 2, 50, 8)
    C[0, 1, 3, 5] = float("nan")
    _, _, status = run(C, [4, 4])
    assert status.tolist() == [0, 1]


def test_fused_cost_matrix_matches_torch_ops_and_gives_the_same_matching():
    """csrc/match_cost.hip against HungarianMatcher.cost_matrix (the reference's op sequence,
    matcher.py:48-88), and forward_many with / without it."""
    from datr_amd import matcher as M
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(11)
    G, B, nq, C = 7, 2, 900, 9
    outs = [{"pred_logits": (torch.randn(B, nq, C, generator=gen) * 2 - 2).to(dev),
             "pred_boxes": torch.cat([torch.rand(B, nq, 2, generator=gen),
                                      torch.rand(B, nq, 2, generator=gen) * 0.5 + 0.01], -1).to(dev)}
            for _ in range(G)]
    targets = [{"labels": torch.randint(1, 9, (n,), generator=gen).to(dev),
                "boxes": torch.cat([torch.rand(n, 2, generator=gen) * 0.6 + 0.2,
                                    torch.rand(n, 2, generator=gen) * 0.2 + 0.05], -1).to(dev)}
               for n in (10, 7)]
    m = M.HungarianMatcher(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal_alpha=0.25)
    logits = torch.cat([o["pred_logits"] for o in outs], 0)
    boxes = torch.cat([o["pred_boxes"] for o in outs], 0)
    ct = m.cost_matrix_transposed(logits, boxes, targets)
    assert bool(m._boxes_ok)
    ref = m.cost_matrix({"pred_logits": logits, "pred_boxes": boxes}, targets)        # [G*B, nq, T]
    torch.testing.assert_close(ct, ref.transpose(1, 2), rtol=1e-5, atol=1e-5)
    fused = m.forward_many(outs, targets)
    M.FUSED_COST = False
    try:
        plain = m.forward_many(outs, targets)
    finally:
        M.FUSED_COST = True
    for a, b in zip(fused, plain):
        for (q1, t1), (q2, t2) in zip(a, b):
            assert torch.equal(q1, q2) and torch.equal(t1, t2)
    # a degenerate box (negative width) clears the validity flag
    boxes[3, 5, 2] = -0.1
    m.cost_matrix_transposed(logits, boxes, targets)
    assert not bool(m._boxes_ok)
