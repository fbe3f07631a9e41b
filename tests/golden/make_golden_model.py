"""Generate tests/golden/model_*.npz by running the REFERENCE model (build-container only).

Imports /root/reference under oracle/ref_shims.py, builds the reference's DINO with the
Cityscapes->Foggy config (`build_dino`, /root/reference/models/dino/dino.py:999), overwrites
its weights with tests/golden/synth.py's deterministic tensors, and records

  model_step.npz   one training forward + criterion + backward at 256x320, B = 1 (one source
                   + one target image): every head output, the DA outputs, the 82-entry loss
                   dict, the weighted total, the Hungarian indices of all 7 matcher calls,
                   per-parameter gradient norms, the running prototypes after the step, and
                   the four random draws of prepare_for_cdn (dn_components.py:64-66,84-85)
  model_step_sim10k.npz   the same step with the Sim10k -> Cityscapes config (num_classes = 2,
                   /root/reference/config/DA/Sim10k2Cityscapes/DINO_4scale_sim2cityscapes.py)
  model_eval.npz   eval-mode forward of the same images + PostProcess(num_select=100)
  model_units.npz  direct calls of the small functions on the path (position embedding,
                   sine query embedding, encoder proposals, focal loss, matcher cost, ...)

    python tests/golden/make_golden_model.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_shims  # noqa: E402

ref_shims.install()
import synth  # noqa: E402
from models.dino.dino import build_dino  # noqa: E402  (the reference's)
from models.dino import utils as ref_utils  # noqa: E402
from models.dino.position_encoding import PositionEmbeddingSineHW as RefPE  # noqa: E402
from models.dino.matcher import HungarianMatcher as RefMatcher  # noqa: E402
from models.dino.DA_utils import get_prototype_class_wise as ref_proto  # noqa: E402
from util.misc import NestedTensor as RefNested, nested_tensor_from_tensor_list as ref_nest  # noqa: E402
from util.misc import inverse_sigmoid as ref_inverse_sigmoid  # noqa: E402
from util import box_ops as ref_box_ops  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def to_np(x):
    return x.detach().cpu().numpy()


class DrawRecorder:
    """Records the outputs of torch.rand_like / torch.randint_like in call order."""

    def __init__(self):
        self.draws = []
        self._rand_like, self._randint_like = torch.rand_like, torch.randint_like

    def __enter__(self):
        def rl(*a, **k):
            out = self._rand_like(*a, **k)
            self.draws.append(out.clone())
            return out

        def ril(*a, **k):
            out = self._randint_like(*a, **k)
            self.draws.append(out.clone())
            return out
        torch.rand_like, torch.randint_like = rl, ril
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.randint_like = self._rand_like, self._randint_like


def flatten_indices(indices_list):
    """list (per matcher call) of list (per image) of (src, tgt) -> int64 array [calls, imgs, 2, T]"""
    return np.stack([np.stack([np.stack([to_np(s), to_np(t)]) for s, t in call])
                     for call in indices_list])


C2F_CONFIG = "config/DA/Cityscapes2FoggyCityscapes/DINO_4scale_C2F.py"
SIM10K_CONFIG = "config/DA/Sim10k2Cityscapes/DINO_4scale_sim2cityscapes.py"     # num_classes = dn_labelbook_size = 2


def training_step(config, num_classes, out_name):
    """One training forward + criterion + backward of the reference model built from `config`;
    writes tests/golden/<out_name> and returns (model, imgs) for the eval leg."""
    args = ref_shims.load_config(config)
    assert args.num_classes == num_classes and args.dn_labelbook_size == num_classes
    torch.manual_seed(0)
    model, criterion, post = build_dino(args)
    synth.synth_init_(model)
    imgs, targets = synth.synth_batch(num_classes=num_classes)

    # ---------------- training step ---------------------------------------------------------
    model.train()
    criterion.train()
    samples = ref_nest(imgs)
    torch.manual_seed(123)
    topk_calls, topk_inputs = [], []
    real_topk = torch.topk

    def rec_topk(inp, k, *a, **kw):
        res = real_topk(inp, k, *a, **kw)
        if k == 900:
            topk_calls.append(res[1].clone())
            topk_inputs.append(inp.detach().clone())      # the scores the selection is made from
        return res
    torch.topk = rec_topk
    try:
        with DrawRecorder() as rec:
            out = model(samples, targets)
    finally:
        torch.topk = real_topk
    assert len(rec.draws) == 4, len(rec.draws)
    assert len(topk_calls) == 2, len(topk_calls)      # source pass, target pass
    loss_dict, indices_list = criterion(out, targets, return_indices=True)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    model.zero_grad()
    total.backward()

    rec_ = {
        "noise_label_p": to_np(rec.draws[0]), "noise_new_label": to_np(rec.draws[1]),
        # the reference draws {0,1} and maps to {-1,+1} afterwards (dn_components.py:84)
        "noise_rand_sign": to_np(rec.draws[2] * 2.0 - 1.0), "noise_rand_part": to_np(rec.draws[3]),
        "topk_source": to_np(topk_calls[0]), "topk_target": to_np(topk_calls[1]),
        "topk_scores_source": to_np(topk_inputs[0]), "topk_scores_target": to_np(topk_inputs[1]),
        "pred_logits": to_np(out["pred_logits"]), "pred_boxes": to_np(out["pred_boxes"]),
        "aux_logits": to_np(torch.stack([a["pred_logits"] for a in out["aux_outputs"]])),
        "aux_boxes": to_np(torch.stack([a["pred_boxes"] for a in out["aux_outputs"]])),
        "interm_logits": to_np(out["interm_outputs"]["pred_logits"]),
        "interm_boxes": to_np(out["interm_outputs"]["pred_boxes"]),
        "init_box_proposal": to_np(out["interm_outputs_for_matching_pre"]["pred_boxes"]),
        "dn_logits": to_np(out["dn_meta"]["output_known_lbs_bboxes"]["pred_logits"]),
        "dn_boxes": to_np(out["dn_meta"]["output_known_lbs_bboxes"]["pred_boxes"]),
        "dn_pad_size": np.int64(out["dn_meta"]["pad_size"]),
        "dn_groups": np.int64(out["dn_meta"]["num_dn_group"]),
        "backbone_DA": to_np(out["da_output"]["backbone_DA"]),
        "da_protos": to_np(out["da_output"]["proto_DA"]["da_protos"]),
        "class_map_source": to_np(out["da_output"]["proto_DA"]["class_map_source"]),
        "class_map_target": to_np(out["da_output"]["proto_DA"]["class_map_target"]),
        "proto_source": to_np(out["da_output"]["global_proto_DA"]["output_source"]),
        "proto_target": to_np(out["da_output"]["global_proto_DA"]["outputs_target"]),
        "global_proto": to_np(model.global_proto), "Amount": to_np(model.Amount),
        "indices": flatten_indices(indices_list),
        "total_loss": to_np(total),
        "loss_keys": np.array(list(loss_dict.keys())),
        "loss_values": np.array([float(v) for v in loss_dict.values()], dtype=np.float64),
        "weight_keys": np.array(list(wd.keys())),
        "weight_values": np.array([float(v) for v in wd.values()], dtype=np.float64),
    }
    # gradient norms keyed by the canonical (smallest) state_dict name of each parameter
    sd = model.state_dict(keep_vars=True)
    names_of = {}
    for k, v in sd.items():
        names_of.setdefault(v.data_ptr(), []).append(k)
    gk, gv = [], []
    for ptr, names in names_of.items():
        p = sd[names[0]]
        if p.requires_grad:
            assert p.grad is not None, names
            gk.append(min(names))
            gv.append(float(p.grad.double().norm()))
    rec_["grad_keys"], rec_["grad_norms"] = np.array(gk), np.array(gv, dtype=np.float64)
    # a few full gradients (small tensors) for a sharper check
    for k in ("class_embed.0.bias", "transformer.level_embed", "label_enc.weight",
              "D_img.classifier.bias", "Proto_D.layers.2.weight",
              "transformer.decoder.layers.5.cross_attn.sampling_offsets.bias",
              "transformer.encoder.layers.0.self_attn.attention_weights.bias",
              "backbone.0.body.layer4.2.conv3.weight"):
        g = sd[k].grad
        rec_["grad::" + k] = to_np(g if g.numel() < 5000 else g.flatten()[:5000])
    rec_["state_keys"] = np.array(list(sd.keys()))
    rec_["state_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, out_name), **rec_)
    print("wrote", out_name, "; total loss", float(total), "keys", len(loss_dict))
    return model, imgs


def main():
    model, imgs = training_step(C2F_CONFIG, 9, "model_step.npz")
    # the Sim10k -> Cityscapes task (README.md:115): one foreground class, so the class heads are [2, 256],
    # label_enc [3, 256], prototypes [2, 256] -- the shapes the per-class kernels see with C = 2
    training_step(SIM10K_CONFIG, 2, "model_step_sim10k.npz")

    # ---------------- eval forward + post-process ---------------------------------------------
    model.eval()
    with torch.no_grad():
        out_e = model(ref_nest(imgs))
        sizes = torch.tensor([[256.0, 320.0], [240.0, 300.0]])
        from models.dino.dino import PostProcess as RefPost
        res = RefPost(num_select=100)(out_e, sizes)
    np.savez_compressed(
        os.path.join(OUT, "model_eval.npz"),
        pred_logits=to_np(out_e["pred_logits"]), pred_boxes=to_np(out_e["pred_boxes"]),
        scores=to_np(torch.stack([r["scores"] for r in res])),
        labels=to_np(torch.stack([r["labels"] for r in res])),
        boxes=to_np(torch.stack([r["boxes"] for r in res])), sizes=to_np(sizes))
    print("wrote model_eval.npz")

    # ---------------- unit-level vectors --------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    u = {}
    mask = torch.zeros(2, 13, 17, dtype=torch.bool)
    mask[1, 10:, :] = True
    mask[1, :, 12:] = True
    pe = RefPE(128, temperatureH=20, temperatureW=20, normalize=True)
    u["pe_mask"] = to_np(mask)
    u["pe_out"] = to_np(pe(RefNested(torch.zeros(2, 4, 13, 17), mask)))
    pos = torch.rand(11, 2, 4, generator=g)
    u["sine_in"], u["sine_out4"] = to_np(pos), to_np(ref_utils.gen_sineembed_for_position(pos))
    u["sine_out2"] = to_np(ref_utils.gen_sineembed_for_position(pos[..., :2]))
    shapes = torch.tensor([[6, 8], [3, 4]])
    mem = torch.randn(2, 60, 16, generator=g)
    pm = torch.zeros(2, 60, dtype=torch.bool)
    pm[1, 40:48] = True
    pm[1, 57:] = True
    om, op = ref_utils.gen_encoder_output_proposals(mem, pm, shapes)
    u["prop_memory"], u["prop_mask"], u["prop_shapes"] = to_np(mem), to_np(pm), to_np(shapes)
    u["prop_out_memory"], u["prop_out"] = to_np(om), to_np(op)
    x = torch.rand(50, generator=g) * 1.2 - 0.1
    u["invsig_in"], u["invsig_out"] = to_np(x), to_np(ref_inverse_sigmoid(x))
    logits = torch.randn(2, 30, 9, generator=g) * 2
    tgt = (torch.rand(2, 30, 9, generator=g) < 0.1).float()
    u["focal_logits"], u["focal_targets"] = to_np(logits), to_np(tgt)
    u["focal_out"] = to_np(ref_utils.sigmoid_focal_loss(logits, tgt, 7.0, alpha=0.25, gamma=2))
    b1 = torch.rand(12, 4, generator=g) * 0.4 + 0.1
    b2 = torch.rand(5, 4, generator=g) * 0.4 + 0.1
    u["boxes1"], u["boxes2"] = to_np(b1), to_np(b2)
    u["giou"] = to_np(ref_box_ops.generalized_box_iou(ref_box_ops.box_cxcywh_to_xyxy(b1),
                                                      ref_box_ops.box_cxcywh_to_xyxy(b2)))
    # matcher: cost matrix is internal, so store inputs + resulting indices
    m_logits = torch.randn(2, 40, 9, generator=g)
    m_boxes = torch.rand(2, 40, 4, generator=g) * 0.4 + 0.1
    m_t = [{"labels": torch.randint(1, 9, (4,), generator=g), "boxes": torch.rand(4, 4, generator=g) * 0.4 + 0.1},
           {"labels": torch.randint(1, 9, (6,), generator=g), "boxes": torch.rand(6, 4, generator=g) * 0.4 + 0.1}]
    idx = RefMatcher(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)({"pred_logits": m_logits, "pred_boxes": m_boxes}, m_t)
    u["match_logits"], u["match_boxes"] = to_np(m_logits), to_np(m_boxes)
    for i, t in enumerate(m_t):
        u[f"match_tlabels{i}"], u[f"match_tboxes{i}"] = to_np(t["labels"]), to_np(t["boxes"])
        u[f"match_src{i}"], u[f"match_tgt{i}"] = to_np(idx[i][0]), to_np(idx[i][1])
    # prototypes: two consecutive calls (running-mean update)
    q = torch.randn(2, 50, 256, generator=g)
    lg = torch.randn(2, 50, 9, generator=g)
    gp, ga = torch.zeros(9, 256), torch.zeros(9)
    p1 = ref_proto(q, lg, 9, global_proto=gp, global_amount=ga)
    q2 = torch.randn(2, 50, 256, generator=g)
    lg2 = torch.randn(2, 50, 9, generator=g)
    p2 = ref_proto(q2, lg2, 9, global_proto=p1[2], global_amount=p1[3])
    u["proto_q"], u["proto_logits"], u["proto_q2"], u["proto_logits2"] = map(to_np, (q, lg, q2, lg2))
    for n, p in (("1", p1), ("2", p2)):
        u["proto_out" + n], u["proto_map" + n] = to_np(p[0]), to_np(p[1])
        u["proto_global" + n], u["proto_amount" + n] = to_np(p[2]), to_np(p[3])
    np.savez_compressed(os.path.join(OUT, "model_units.npz"), **u)
    print("wrote model_units.npz")


if __name__ == "__main__":
    main()
