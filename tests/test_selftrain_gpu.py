"""GPU: the teacher-student step on the HIP path against one step of the reference's own
`train_one_epoch_with_self_training` (tests/golden/selftrain_step.npz, BASELINE config 5).
The three top-900 selections (teacher, student source, student target) are substituted by the
reference's, as in test_model_gpu.py: near-tied encoder scores may swap ranks between devices."""
import numpy as np
import pytest
import torch

from helpers import build_model, load_npz, t
from test_selftrain_cpu import batch

pytestmark = pytest.mark.gpu


def test_teacher_student_step_on_device():
    from datr_amd.config import get_param_dict
    from datr_amd.ema import ModelEMA
    from datr_amd.engine import train_one_epoch_with_self_training
    dev = torch.device("cuda:0")
    g = load_npz("selftrain_step.npz")
    args, model, criterion, _ = build_model()
    args.pseudo_label_threshold = float(g["threshold"])
    model.to(dev)
    criterion.to(dev)
    model.dn_noise_override = {
        "label_p": t(g["noise_label_p"]), "new_label": t(g["noise_new_label"]),
        "rand_sign": t(g["noise_rand_sign"]), "rand_part": t(g["noise_rand_part"])}
    teacher = ModelEMA(model, decay=args.ema_decay_teacher)
    student_calls = [t(g["topk_source"]).to(dev), t(g["topk_target"]).to(dev)]
    state = {"i": 0}

    def student_selection(scores):
        if scores.shape[0] == student_calls[0].shape[0] + student_calls[1].shape[0]:
            return torch.cat(student_calls, 0)         # merged source + target decoder pass
        idx = student_calls[state["i"] % 2]
        state["i"] += 1
        return idx
    model.transformer.select_queries = student_selection
    teacher.ema.transformer.select_queries = lambda scores: t(g["topk_teacher"]).to(dev)
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr, weight_decay=args.weight_decay)
    loader = batch()
    stats = train_one_epoch_with_self_training(model, teacher, criterion, loader, loader, optimizer,
                                               dev, 0, args.clip_max_norm, args=args)
    last = stats["_last"]
    pt = last["pseudo_targets"]
    assert len(pt) == 1
    assert torch.equal(pt[0]["labels"].cpu(), t(g["pseudo_labels"]))          # index selection: exact
    torch.testing.assert_close(pt[0]["boxes"].cpu(), t(g["pseudo_boxes"]), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(pt[0]["scores"].cpu(), t(g["pseudo_scores"]), rtol=1e-3, atol=1e-3)
    ref = {str(k): float(v) for k, v in zip(g["stat_keys"], g["stat_values"])}
    for k, v in ref.items():
        assert k in stats, k
        if k == "class_error":
            continue                      # a count over 5 boxes: one near-tie flips it by 20
        assert abs(stats[k] - v) <= 5e-3 * abs(v) + 1e-4, (k, stats[k], v)
    assert last["num_pseudo_images"] == 1 and last["loss_self_training_sum"] > 0
    sd = model.state_dict()
    norms = np.array([float(sd[str(k)].double().norm()) for k in g["param_keys"]])
    np.testing.assert_allclose(norms, g["param_norms"], rtol=5e-5, atol=1e-7)


@pytest.mark.parametrize("n,classes", [(1, 1), (7, 2), (100, 8), (900, 3)])
def test_device_nms_keeps_exactly_the_host_indices(n, classes):
    """csrc/nms.hip (rank + class offsets + IoU + greedy scan in one launch) against the float32 host
    formulation of torchvision's batched_nms (self_training.nms_host): identical kept indices in
    identical order, on crowded boxes (clusters of near-duplicates), tied scores and IoUs sitting
    next to the 0.7 threshold."""
    from datr_amd.self_training import batched_nms
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    centres = torch.rand(max(n // 6, 1), 2, generator=g) * 900 + 100
    pick = torch.randint(0, centres.shape[0], (n,), generator=g)
    cxcy = centres[pick] + torch.randn(n, 2, generator=g) * 6
    wh = 80 + torch.rand(n, 2, generator=g) * 40
    boxes = torch.cat([cxcy - wh / 2, cxcy + wh / 2], 1)
    scores = (torch.rand(n, generator=g) * 20).round() / 20          # many exact ties
    labels = torch.randint(1, classes + 1, (n,), generator=g)
    for thr in (0.7, 0.5):
        host = batched_nms(boxes, scores, labels, thr)
        mine = batched_nms(boxes.to(dev), scores.to(dev), labels.to(dev), thr)
        assert mine.dtype == torch.int64 and torch.equal(mine.cpu(), host)
    if n >= 100:
        assert 0 < host.numel() < n


def test_device_nms_of_no_boxes_is_empty():
    from datr_amd.self_training import batched_nms
    dev = torch.device("cuda:0")
    out = batched_nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev),
                      torch.zeros(0, dtype=torch.int64, device=dev), 0.7)
    assert out.shape == (0,) and out.dtype == torch.int64


def test_ema_update_kernel_is_the_key_walk_bit_for_bit():
    """csrc/ema.hip (one launch over all tensors, aliased heads replayed 12 times in registers)
    against the reference's loop `v *= d; v += (1. - d) * msd[k]` over the state_dict keys
    (/root/reference/models/dino/EMA.py:46-50) run on the device with torch ops: identical bits, for
    ModelEMA's ramped decay and CosineEMA's, over several updates; the tables survive a second model
    and are rebuilt when storages move."""
    import copy
    from datr_amd import ema as ema_mod
    from datr_amd.ema import CosineEMA, ModelEMA
    from tests.helpers import build_model
    dev = torch.device("cuda:0")
    _, model, _, _ = build_model()
    model.to(dev)
    for teacher, decay in ((ModelEMA(model, decay=0.9996, updates=3000), None), (CosineEMA(model, 0.99, 0.9999, 10), 0.995)):
        shadow = copy.deepcopy(teacher.ema)
        calls = []
        real = ema_mod._DevicePlan.run
        ema_mod._DevicePlan.run = lambda self, d: (calls.append(self.npieces), real(self, d))[1]
        try:
            for it in range(3):
                with torch.no_grad():
                    for p in model.parameters():
                        p.add_(0.03 * torch.randn_like(p))
                d = teacher.decay(teacher.updates + 1) if decay is None else decay
                if decay is not None:
                    teacher.decay = decay
                msd = model.state_dict()
                with torch.no_grad():
                    for k, v in shadow.state_dict().items():
                        if v.dtype.is_floating_point:
                            v *= d
                            v += (1.0 - d) * msd[k].detach()
                teacher.update(model)
                for (k, mine), (_, ref) in zip(teacher.ema.state_dict().items(), shadow.state_dict().items()):
                    assert torch.equal(mine, ref), (it, k)
        finally:
            ema_mod._DevicePlan.run = real
        assert len(calls) == 3 and calls[0] > 10000            # one launch per update, ~49.6 M elements / 4096
    # a storage that moved (load_state_dict keeps storages; .data = ... does not): the plan follows
    teacher = ModelEMA(model, decay=0.999, updates=10)
    teacher.update(model)
    plan = ema_mod._PLANS[id(teacher.ema)]
    w = model.transformer.level_embed
    w.data = w.data.clone()
    teacher.update(model)
    assert ema_mod._PLANS[id(teacher.ema)] is not plan
