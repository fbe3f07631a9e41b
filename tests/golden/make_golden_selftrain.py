"""Generate tests/golden/selftrain_step.npz by running the REFERENCE's teacher-student epoch
function for one step (build container only).

Calls /root/reference/engine.py `train_one_epoch_with_self_training` (:146-342) with a
one-batch loader: EMA teacher = `ModelEMA(student)` (/root/reference/models/dino/EMA.py:21),
student with synthetic weights (tests/golden/synth.py), weak / strong target images.  Records
the pseudo labels the teacher produced, the CDN draws, the top-900 selections, the returned
stats and per-parameter norms after the optimizer step.

    python tests/golden/make_golden_selftrain.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_shims  # noqa: E402

ref_shims.install()
import synth  # noqa: E402
import engine as ref_engine  # noqa: E402  (the reference's)
from models.dino import self_training_utils as ref_st  # noqa: E402
from models.dino.dino import build_dino  # noqa: E402
from models.dino.EMA import ModelEMA  # noqa: E402
from util.get_param_dicts import get_param_dict  # noqa: E402
from util.misc import nested_tensor_from_tensor_list as ref_nest  # noqa: E402

from make_golden_model import DrawRecorder, to_np  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
THRESHOLD = 0.02        # synthetic weights give low scores; the C2F config uses 0.3


def strong_batch():
    imgs, targets = synth.synth_batch()
    g = torch.Generator().manual_seed(99)
    strong = [imgs[0], imgs[1] + 0.3 * torch.randn(imgs[1].shape, generator=g)]
    tgt_meta = [{"image_id": torch.tensor([7]), "area": torch.tensor([1.0]),
                 "iscrowd": torch.tensor([0]), "orig_size": torch.tensor([480, 600]),
                 "size": torch.tensor([240, 300]), "boxes": torch.zeros(0, 4),
                 "labels": torch.zeros(0, dtype=torch.long)}]
    return imgs, strong, targets, tgt_meta


def main():
    tmp = tempfile.mkdtemp()
    args = ref_shims.load_config(pseudo_label_threshold=THRESHOLD, output_dir=tmp,
                                 param_dict_type="default")
    torch.manual_seed(0)
    model, criterion, _ = build_dino(args)
    synth.synth_init_(model)
    teacher = ModelEMA(model, decay=args.ema_decay_teacher)
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr,
                                  weight_decay=args.weight_decay)
    imgs, strong, targets, tgt_meta = strong_batch()
    loader = [(ref_nest(imgs), tuple(targets), tuple(tgt_meta), ref_nest(strong))]

    captured = {}
    orig_rescale = ref_engine.rescale_pseudo_targets

    def rec_rescale(img, pseudo, *a, **k):
        out = orig_rescale(img, pseudo, *a, **k)
        captured["pseudo"] = {i: {kk: vv.clone() for kk, vv in t.items()} for i, t in out.items()}
        return out
    ref_engine.rescale_pseudo_targets = rec_rescale

    topk_calls = []
    real_topk = torch.topk

    def rec_topk(inp, k, *a, **kw):
        res = real_topk(inp, k, *a, **kw)
        if k == 900:
            topk_calls.append(res[1].clone())
        return res
    torch.topk = rec_topk
    torch.manual_seed(321)
    try:
        with DrawRecorder() as rec:
            stats = ref_engine.train_one_epoch_with_self_training(
                model, teacher, criterion, loader, loader, optimizer, torch.device("cpu"), 0,
                args.clip_max_norm, wo_class_error=False, lr_scheduler=None, args=args)
    finally:
        torch.topk = real_topk
        ref_engine.rescale_pseudo_targets = orig_rescale
    assert len(rec.draws) == 4 and len(topk_calls) == 3, (len(rec.draws), len(topk_calls))
    pseudo = captured["pseudo"]
    assert 0 in pseudo and len(pseudo[0]["labels"]) > 0, "threshold gave no pseudo labels"

    sd = model.state_dict()
    names_of = {}
    for k, v in sd.items():
        names_of.setdefault(v.data_ptr(), []).append(k)
    keys = sorted(min(n) for n in names_of.values())
    out = {
        "threshold": np.float64(THRESHOLD),
        "noise_label_p": to_np(rec.draws[0]), "noise_new_label": to_np(rec.draws[1]),
        "noise_rand_sign": to_np(rec.draws[2] * 2.0 - 1.0), "noise_rand_part": to_np(rec.draws[3]),
        "topk_teacher": to_np(topk_calls[0]), "topk_source": to_np(topk_calls[1]),
        "topk_target": to_np(topk_calls[2]),
        "pseudo_labels": to_np(pseudo[0]["labels"]), "pseudo_boxes": to_np(pseudo[0]["boxes"]),
        "pseudo_scores": to_np(pseudo[0]["scores"]),
        "stat_keys": np.array(sorted(stats.keys())),
        "stat_values": np.array([float(stats[k]) for k in sorted(stats.keys())], dtype=np.float64),
        "param_keys": np.array(keys),
        "param_norms": np.array([float(sd[k].double().norm()) for k in keys], dtype=np.float64),
    }
    np.savez_compressed(os.path.join(OUT, "selftrain_step.npz"), **out)
    print("wrote selftrain_step.npz:", len(pseudo[0]["labels"]), "pseudo labels; loss", stats["loss"])


if __name__ == "__main__":
    main()
