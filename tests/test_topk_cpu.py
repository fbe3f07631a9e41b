"""The query selection as a function (datr_amd.fused.topk_rows; SURVEY.md 8 row a5, bit-exact
indices): against the reference's own selection recorded by tests/golden/make_golden_model.py
(`topk_scores_*` = the tensor the reference hands to torch.topk, deformable_transformer.py:342;
`topk_*` = the indices it got), and the defined order on ties."""
import numpy as np
import torch

from helpers import load_npz, t


def expected_order(x, k):
    """descending value, NaN first, equal values by ascending index -- written out"""
    out = []
    for row in x.tolist():
        order = sorted(range(len(row)), key=lambda j: ((0, 0.0) if row[j] != row[j] else (1, -row[j]), j))
        out.append(order[:k])
    return torch.tensor(out)


def check_against_reference_selection(select, g, name):
    scores, ref = t(g[f"topk_scores_{name}"]), t(g[f"topk_{name}"])
    mine = select(scores)
    assert mine.shape == ref.shape and mine.dtype == torch.int64
    vm, vr = scores.gather(1, mine.cpu()), scores.gather(1, ref)
    assert torch.equal(vm, vr), "the selected SCORES differ from the reference's"
    for r in range(scores.shape[0]):
        vals, counts = torch.unique(vr[r], return_counts=True)
        unique_pos = torch.isin(vr[r], vals[counts == 1])
        # distinct scores: the reference's index, bit-exact
        assert torch.equal(mine[r].cpu()[unique_pos], ref[r][unique_pos])
        # tied scores (torch.topk leaves their order open): the same tokens, lowest index first
        for v in vals[counts > 1]:
            pos = vr[r] == v
            assert sorted(mine[r].cpu()[pos].tolist()) == sorted(ref[r][pos].tolist())
            assert mine[r].cpu()[pos].tolist() == sorted(mine[r].cpu()[pos].tolist())
    return int((~unique_pos).sum())


def test_selection_equals_the_reference_on_its_own_scores():
    from datr_amd.fused import topk_rows
    g = load_npz("model_step.npz")
    tied_s = check_against_reference_selection(lambda s: topk_rows(s, 900)[1], g, "source")
    tied_t = check_against_reference_selection(lambda s: topk_rows(s, 900)[1], g, "target")
    assert tied_s == 0            # source scores are distinct: the whole selection is bit-exact
    assert tied_t > 0             # the padded target image produces tied tokens: exercised too


def test_defined_order_on_ties_nan_and_inf():
    from datr_amd.fused import topk_rows
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 40, (3, 3000), generator=g).float()
    x[0, 5] = float("nan"); x[0, 2000] = float("nan"); x[1, 17] = float("inf"); x[2, 3] = float("-inf")
    v, i = topk_rows(x, 900)
    assert torch.equal(i, expected_order(x, 900))
    assert torch.equal(torch.nan_to_num(v, nan=123.0), torch.nan_to_num(x.gather(1, i), nan=123.0))
    # distinct scores: torch.topk itself
    y = torch.randn(2, 5000, generator=g)
    assert torch.equal(topk_rows(y, 300)[1], torch.topk(y, 300, dim=1)[1])
