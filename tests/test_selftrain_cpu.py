"""CPU: the teacher-student step (datr_amd.engine.train_one_epoch_with_self_training,
datr_amd.ema, datr_amd.self_training) against one step of the reference's own epoch function
(tests/golden/make_golden_selftrain.py -> selftrain_step.npz)."""
import argparse

import numpy as np
import torch

from helpers import build_model, load_npz, patch_msda_with_oracle, t


def batch():
    import synth
    from datr_amd.nested import nested_tensor_from_tensor_list
    imgs, targets = synth.synth_batch()
    g = torch.Generator().manual_seed(99)
    strong = [imgs[0], imgs[1] + 0.3 * torch.randn(imgs[1].shape, generator=g)]
    tgt_meta = [{"image_id": torch.tensor([7]), "area": torch.tensor([1.0]),
                 "iscrowd": torch.tensor([0]), "orig_size": torch.tensor([480, 600]),
                 "size": torch.tensor([240, 300]), "boxes": torch.zeros(0, 4),
                 "labels": torch.zeros(0, dtype=torch.long)}]
    return [(nested_tensor_from_tensor_list(imgs), tuple(targets), tuple(tgt_meta),
             nested_tensor_from_tensor_list(strong))]


def test_teacher_student_step_matches_reference(monkeypatch):
    from datr_amd.config import get_param_dict
    from datr_amd.ema import ModelEMA
    from datr_amd.engine import train_one_epoch_with_self_training
    patch_msda_with_oracle(monkeypatch, kind="grid_sample")
    g = load_npz("selftrain_step.npz")
    args, model, criterion, _ = build_model()
    args.pseudo_label_threshold = float(g["threshold"])
    model.merge_encoder_passes = False                 # the reference's call structure
    model.dn_noise_override = {
        "label_p": t(g["noise_label_p"]), "new_label": t(g["noise_new_label"]),
        "rand_sign": t(g["noise_rand_sign"]), "rand_part": t(g["noise_rand_part"])}
    teacher = ModelEMA(model, decay=args.ema_decay_teacher)
    assert not teacher.ema.training and not any(p.requires_grad for p in teacher.ema.parameters())
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr, weight_decay=args.weight_decay)
    loader = batch()
    stats = train_one_epoch_with_self_training(model, teacher, criterion, loader, loader, optimizer,
                                               torch.device("cpu"), 0, args.clip_max_norm, args=args)
    last = stats["_last"]
    # pseudo labels produced by the teacher: classes bit-exact, boxes / scores to rounding
    pt = last["pseudo_targets"]
    assert len(pt) == 1
    assert torch.equal(pt[0]["labels"], t(g["pseudo_labels"]))
    torch.testing.assert_close(pt[0]["boxes"], t(g["pseudo_boxes"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pt[0]["scores"], t(g["pseudo_scores"]), rtol=1e-5, atol=1e-6)
    # every stat the reference's MetricLogger averaged (loss, scaled and unscaled terms, lr, class_error)
    ref = {str(k): float(v) for k, v in zip(g["stat_keys"], g["stat_values"])}
    for k, v in ref.items():
        assert k in stats, k
        assert abs(stats[k] - v) <= 1e-4 * abs(v) + 1e-5, (k, stats[k], v)
    assert last["num_pseudo_images"] == 1 and last["loss_self_training_sum"] > 0
    # parameters after clip + AdamW step
    sd = model.state_dict()
    norms = np.array([float(sd[str(k)].double().norm()) for k in g["param_keys"]])
    np.testing.assert_allclose(norms, g["param_norms"], rtol=1e-6, atol=1e-9)


def test_ema_update_rule():
    from datr_amd.ema import CosineEMA, ModelEMA
    m = torch.nn.Linear(4, 3)
    ema = ModelEMA(m, decay=0.9997)
    w0 = ema.ema.weight.clone()
    with torch.no_grad():
        m.weight.add_(1.0)
    ema.update(m)
    d = 0.9997 * (1 - np.exp(-1 / 2000))            # EMA.py:37
    assert torch.equal(ema.ema.weight, w0 * d + (1 - d) * m.weight.detach())
    c = CosineEMA(m, decay_start=0.9, decay_end=0.99, total_epoch=10)
    c.update_decay(5)
    assert abs(c.decay - (0.99 - 0.09 * (np.cos(np.pi * 0.5) + 1) / 2)) < 1e-12


def test_no_pseudo_labels_keeps_the_step_alive(monkeypatch):
    """Threshold above every score: the target criterion returns {} and the step still runs
    (/root/reference/models/dino/dino.py:761-774, engine.py:253-255)."""
    from datr_amd.config import get_param_dict
    from datr_amd.ema import ModelEMA
    from datr_amd.engine import train_one_epoch_with_self_training
    patch_msda_with_oracle(monkeypatch, kind="c")
    args, model, criterion, _ = build_model()
    args.pseudo_label_threshold = 0.999
    teacher = ModelEMA(model)
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr, weight_decay=args.weight_decay)
    loader = batch()
    stats = train_one_epoch_with_self_training(model, teacher, criterion, loader, None, optimizer,
                                               torch.device("cpu"), 0, args.clip_max_norm, args=args)
    assert stats["_last"]["num_pseudo_images"] == 0 and stats["_last"]["target_loss_dict"] == {}
    assert np.isfinite(stats["loss"])


def test_ema_update_of_aliased_heads_follows_the_per_key_loop():
    """the reference updates once per state_dict KEY (EMA.py:46-50); the detector's
    shared heads sit under 12 keys each, so their effective decay is d^12.  Checked on the real
    detector against the per-key loop written out."""
    from datr_amd.ema import ModelEMA
    args, model, _, _ = build_model()
    teacher = ModelEMA(model, decay=0.9996, updates=5000)
    expect = {k: v.clone() for k, v in teacher.ema.state_dict().items()}
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    d = teacher.decay(5001)
    msd = model.state_dict()
    # the reference's loop on a private copy that keeps the aliasing
    import copy
    shadow = copy.deepcopy(teacher.ema)
    with torch.no_grad():
        for k, v in shadow.state_dict().items():
            if v.dtype.is_floating_point:
                v *= d
                v += (1.0 - d) * msd[k].detach()
    teacher.update(model)
    aliases = {}
    for k, v in teacher.ema.state_dict().items():
        aliases.setdefault(v.data_ptr(), []).append(k)
    assert max(len(v) for v in aliases.values()) == 12
    for (k, mine), (_, ref) in zip(teacher.ema.state_dict().items(), shadow.state_dict().items()):
        assert torch.equal(mine, ref), k                      # the same arithmetic: bit for bit
    # and the aliased heads did move (1 - d^12) / (1 - d) ~ 7.8x further than a single update would have
    k = "class_embed.0.weight"
    single = expect[k] * d + (1 - d) * msd[k]
    assert float((teacher.ema.state_dict()[k] - expect[k]).norm()) > 5 * float((single - expect[k]).norm())


def test_host_nms_hand_computed_case():
    """Three overlapping boxes of one class and one of another: IoU(0,1) = 0.81 > 0.7 suppresses the
    lower-scored of the two, IoU(0,2) = 0.43 does not; the other class is never touched."""
    from datr_amd.self_training import batched_nms
    boxes = torch.tensor([[0., 0., 10., 10.], [1., 0., 11., 10.], [4., 0., 14., 10.], [0., 0., 10., 10.]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6])
    labels = torch.tensor([1, 1, 1, 2])
    assert batched_nms(boxes, scores, labels, 0.7).tolist() == [0, 2, 3]
    assert batched_nms(boxes, scores, labels, 0.4).tolist() == [0, 3]
    assert batched_nms(boxes, torch.tensor([0.5, 0.5, 0.5, 0.5]), labels, 0.7).tolist() == [0, 2, 3]
