"""CPU: pin the focal-loss oracle (oracle/focal_ref.c, oracle/focal_oracle.py) against the value
the reference's own `sigmoid_focal_loss` produced (tests/golden/model_units.npz: focal_*)."""
import torch

from helpers import load_npz, t
from oracle import focal_oracle as FO


def golden():
    u = load_npz("model_units.npz")
    logits, onehot = t(u["focal_logits"]), t(u["focal_targets"])     # [2,30,9] each
    # the golden call was sigmoid_focal_loss(logits, targets, num_boxes=7) = loss.mean(1).sum()/7
    return logits, onehot, float(u["focal_out"])


def test_torch_formulation_matches_reference_value():
    logits, onehot, ref = golden()
    # the golden targets are a random multi-hot mask, so evaluate element-wise with that mask
    prob = logits.sigmoid()
    ce = torch.nn.functional.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
    p_t = prob * onehot + (1 - prob) * (1 - onehot)
    loss = (0.25 * onehot + 0.75 * (1 - onehot)) * ce * (1 - p_t) ** 2
    torch.testing.assert_close(float(loss.mean(1).sum() / 7.0), ref, rtol=1e-6, atol=1e-7)


def test_c_oracle_equals_torch_formulation_on_index_targets():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(3, 50, 9, generator=g) * 3
    target = torch.randint(0, 10, (3, 50), generator=g)          # 9 == no object
    for alpha, gamma in ((0.25, 2.0), (-1.0, 2.0), (0.5, 1.5)):
        a = FO.focal_sums_c(logits, target, alpha, gamma)
        b = FO.focal_sums_torch(logits, target, alpha, gamma).double()
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


def test_index_form_equals_reference_normalisation():
    """loss.mean(1).sum() / num_boxes * Q  ==  sum / num_boxes  (dino.py:526, utils.py:104)."""
    from datr_amd.criterion import sigmoid_focal_loss
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(2, 40, 9, generator=g)
    target = torch.randint(0, 10, (2, 40), generator=g)
    onehot = torch.zeros(2, 40, 10).scatter_(2, target.unsqueeze(-1), 1)[..., :9]
    ref = sigmoid_focal_loss(logits, onehot, 5.0, alpha=0.25, gamma=2) * 40
    mine = FO.focal_sums_torch(logits.view(1, 80, 9), target.view(1, 80))[0] / 40 / 5.0 * 40
    torch.testing.assert_close(mine, ref, rtol=1e-6, atol=1e-7)
