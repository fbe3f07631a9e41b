"""Device top-k (csrc/topk.hip through the C ABI) == the defined order, == the reference's
selection on the reference's own scores (tests/golden/model_step.npz), incl. tie-heavy rows."""
import pytest
import torch

from helpers import load_npz
from test_topk_cpu import check_against_reference_selection, expected_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_device_selection_equals_the_reference_on_its_own_scores(dev):
    from datr_amd.fused import topk_rows
    g = load_npz("model_step.npz")
    sel = lambda s: topk_rows(s.to(dev), 900)[1]
    assert check_against_reference_selection(sel, g, "source") == 0
    assert check_against_reference_selection(sel, g, "target") > 0


def test_transformer_select_queries_uses_it(dev):
    from datr_amd.transformer import DeformableTransformer
    tr = DeformableTransformer.__new__(DeformableTransformer)
    tr.num_queries = 900
    g = load_npz("model_step.npz")
    s = torch.from_numpy(g["topk_scores_source"]).to(dev)
    assert torch.equal(tr.select_queries(s).cpu(), torch.from_numpy(g["topk_source"]))


@pytest.mark.parametrize("rows,n,k,levels", [
    (4, 22223, 900, 12),        # the merged encoder pass, tie-heavy: 12 distinct values
    (2, 22223, 900, 0),         # distinct scores
    (2, 1700, 900, 3),
    (3, 8100, 300, 50),         # PostProcess: 900 queries x 9 classes, num_select 300
    (2, 1024, 1024, 5),         # k == n == the kernel's maximum
    (1, 5000, 1, 2),
    (5, 70000, 1000, 7),        # longer than any LDS-resident row
])
def test_defined_order(dev, rows, n, k, levels):
    from datr_amd.fused import topk_rows
    g = torch.Generator().manual_seed(n + k)
    x = torch.randint(0, levels, (rows, n), generator=g).float() if levels else torch.randn(rows, n, generator=g)
    v, i = topk_rows(x.to(dev), k)
    assert torch.equal(i.cpu(), expected_order(x, k))
    assert torch.equal(v.cpu(), x.gather(1, i.cpu()))
    if not levels:
        assert torch.equal(i.cpu(), torch.topk(x, k, dim=1)[1])
    # reproducible launch to launch
    assert torch.equal(topk_rows(x.to(dev), k)[1], i)


def test_nan_inf_and_errors(dev):
    from datr_amd.fused import topk_rows
    x = torch.randn(2, 4000)
    x[0, 5] = float("nan"); x[0, 3000] = float("nan"); x[1, 7] = float("inf"); x[1, 9] = float("-inf")
    i = topk_rows(x.to(dev), 900)[1].cpu()
    assert torch.equal(i, expected_order(x, 900))
    with pytest.raises(NotImplementedError):
        topk_rows(x.to(dev), 1025)
    # gradients flow through the returned values (PostProcess does not need them; the API allows it)
    y = torch.randn(2, 100, device=dev, requires_grad=True)
    topk_rows(y, 10)[0].sum().backward()
    assert int((y.grad != 0).sum()) == 20
