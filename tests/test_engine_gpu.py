"""GPU: two optimizer steps of `datr_amd.engine.train_one_epoch` on the HIP path (merged
source+target passes, fused kernels, device Hungarian) against the reference's own
`engine.train_one_epoch` (tests/golden/engine_epoch.npz).
The top-900 selections are the reference's at both steps (as in test_model_gpu.py: scores tied
to the last bit may swap ranks between devices); the build's own selection must still pick the
same tokens at step 0 up to such swaps."""
import pytest
import torch

from helpers import build_model, load_npz, t
from test_engine_cpu import EpochProbe

pytestmark = pytest.mark.gpu


def test_two_burn_in_steps_on_device():
    from datr_amd.config import get_param_dict
    from datr_amd.engine import train_one_epoch
    dev = torch.device("cuda:0")
    g = load_npz("engine_epoch.npz")
    args, model, criterion, _ = build_model()
    model.to(dev)
    criterion.to(dev)
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr, weight_decay=args.weight_decay)
    probe = EpochProbe(model, criterion, optimizer, g, dev, force_from_step=0)
    stats = train_one_epoch(model, criterion, probe.loader(), optimizer, dev, 0, args.clip_max_norm,
                            args=args)
    assert len(probe.losses) == 2 and len(probe.deltas) == 2
    # own selection at step 0: at most a handful of rank swaps / boundary ties against the CPU run
    mine, ref = probe.build_selection(0), probe.reference_selection(0)
    common = [len(set(a.tolist()) & set(b.tolist())) for a, b in zip(mine, ref)]
    assert min(common) >= ref.shape[1] - 5, common
    # losses within the north-star tolerance (1e-3 on what follows the logits); parameter changes
    # after clip + AdamW: the first step is ~ -lr * sign(grad), so a gradient element at rounding
    # level may flip a whole lr -- norms of the changes still agree to a percent
    probe.check_step(0, loss_rtol=2e-3, delta_rtol=1e-2, skip_counts=True)
    probe.check_step(1, loss_rtol=5e-3, delta_rtol=3e-2, skip_counts=True)
    probe.check_final(stats, stat_rtol=5e-3, norm_rtol=1e-4, delta_cos=0.995, skip_counts=True)
    torch.testing.assert_close(model.global_proto.cpu(), t(g["global_proto"]), rtol=5e-3, atol=2e-3)
    torch.testing.assert_close(model.Amount.cpu(), t(g["Amount"]), rtol=0, atol=3.0)
