"""Generate tests/golden/engine_epoch.npz by running the REFERENCE's burn-in epoch function for
two optimizer steps (build container only).

Calls /root/reference/engine.py `train_one_epoch` (:29-142) with a two-batch loader (two
different synthetic image pairs), the reference's parameter groups
(/root/reference/util/get_param_dicts.py:23-31), AdamW and clip_max_norm from the
Cityscapes->Foggy config.  Records, per step, the CDN draws and the top-900 selections; the
loss dict of every criterion call and every parameter's cumulative change after each optimizer
step; the stats the epoch function returns (MetricLogger global averages over both steps); and
after BOTH steps the norm of every parameter, the norm of its total change, a few full deltas and the
running prototype state -- SURVEY.md section 8 row a17.

    python tests/golden/make_golden_engine.py
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
from models.dino.dino import build_dino  # noqa: E402
from util.get_param_dicts import get_param_dict  # noqa: E402
from util.misc import nested_tensor_from_tensor_list as ref_nest  # noqa: E402

from make_golden_model import DrawRecorder, to_np  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
STEPS = 2
FULL_DELTAS = ("class_embed.0.bias", "transformer.level_embed", "D_img.classifier.bias",
               "transformer.decoder.layers.5.cross_attn.sampling_offsets.bias",
               "transformer.encoder.layers.0.norm1.weight", "bbox_embed.0.layers.2.bias")


def batches():
    """Step s uses synth_batch(seed = 1 + s): different images, boxes and labels per step."""
    return [synth.synth_batch(seed=1 + s) for s in range(STEPS)]


def main():
    tmp = tempfile.mkdtemp()
    args = ref_shims.load_config(output_dir=tmp, param_dict_type="default")
    torch.manual_seed(0)
    model, criterion, _ = build_dino(args)
    synth.synth_init_(model)
    optimizer = torch.optim.AdamW(get_param_dict(args, model), lr=args.lr,
                                  weight_decay=args.weight_decay)
    loader = [(ref_nest(imgs), tuple(targets), None, None) for imgs, targets in batches()]

    sd = model.state_dict()
    names_of = {}
    for k, v in sd.items():
        names_of.setdefault(v.data_ptr(), []).append(k)
    keys = sorted(min(n) for n in names_of.values())
    before = {k: sd[k].detach().clone() for k in keys}

    topk_calls = []
    real_topk = torch.topk

    def rec_topk(inp, k, *a, **kw):
        res = real_topk(inp, k, *a, **kw)
        if k == 900:
            topk_calls.append(res[1].clone())
        return res
    torch.topk = rec_topk
    # per-step captures: the loss dict of every criterion call, the parameter change after
    # every optimizer step
    step_losses, step_deltas = [], []
    hook = criterion.register_forward_hook(
        lambda mod, inp, res: step_losses.append({k: float(v) for k, v in res.items()}))
    real_step = optimizer.step

    def rec_step(*a, **kw):
        res = real_step(*a, **kw)
        cur = model.state_dict()
        step_deltas.append(np.array([float((cur[k].double() - before[k].double()).norm())
                                     for k in keys], dtype=np.float64))
        return res
    optimizer.step = rec_step
    torch.manual_seed(777)
    try:
        with DrawRecorder() as rec:
            stats = ref_engine.train_one_epoch(
                model, criterion, loader, optimizer, torch.device("cpu"), 0, args.clip_max_norm,
                wo_class_error=False, lr_scheduler=None, args=args)
    finally:
        torch.topk = real_topk
        hook.remove()
    assert len(step_losses) == STEPS and len(step_deltas) == STEPS
    assert len(rec.draws) == 4 * STEPS and len(topk_calls) == 2 * STEPS, \
        (len(rec.draws), len(topk_calls))

    sd = model.state_dict()
    out = {
        "steps": np.int64(STEPS), "clip_max_norm": np.float64(args.clip_max_norm),
        "lr": np.float64(args.lr), "lr_backbone": np.float64(args.lr_backbone),
        "weight_decay": np.float64(args.weight_decay),
        "stat_keys": np.array(sorted(stats.keys())),
        "stat_values": np.array([float(stats[k]) for k in sorted(stats.keys())], dtype=np.float64),
        "param_keys": np.array(keys),
        "param_norms": np.array([float(sd[k].double().norm()) for k in keys], dtype=np.float64),
        "delta_norms": np.array([float((sd[k].double() - before[k].double()).norm()) for k in keys],
                                dtype=np.float64),
        "global_proto": to_np(model.global_proto), "Amount": to_np(model.Amount),
    }
    for s in range(STEPS):
        d = rec.draws[4 * s:4 * s + 4]
        out[f"step{s}/noise_label_p"] = to_np(d[0])
        out[f"step{s}/noise_new_label"] = to_np(d[1])
        out[f"step{s}/noise_rand_sign"] = to_np(d[2] * 2.0 - 1.0)     # dn_components.py:84
        out[f"step{s}/noise_rand_part"] = to_np(d[3])
        out[f"step{s}/topk_source"] = to_np(topk_calls[2 * s])
        out[f"step{s}/topk_target"] = to_np(topk_calls[2 * s + 1])
        out[f"step{s}/loss_keys"] = np.array(list(step_losses[s].keys()))
        out[f"step{s}/loss_values"] = np.array(list(step_losses[s].values()), dtype=np.float64)
        out[f"step{s}/delta_norms"] = step_deltas[s]          # cumulative change after step s
    for k in FULL_DELTAS:
        out["delta::" + k] = to_np(sd[k] - before[k])
    np.savez_compressed(os.path.join(OUT, "engine_epoch.npz"), **out)
    moved = int((out["delta_norms"] > 0).sum())
    print(f"wrote engine_epoch.npz: loss {stats['loss']:.6f}, {moved}/{len(keys)} tensors moved")


if __name__ == "__main__":
    main()
