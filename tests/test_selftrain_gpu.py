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
