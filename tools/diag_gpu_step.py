"""Diagnostic (GPU): where does the HIP-path training step deviate from the golden step?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_model, load_npz, run_training_step, t
g = load_npz("model_step.npz")
dev = torch.device("cuda:0")
_, model, criterion, _ = build_model("cuda:0")
out, loss_dict, indices_list, total = run_training_step(model, criterion, dev, g)
def q(name, a, b):
    d = (a.float().cpu() - t(b)).abs()
    print(f"{name:28s} max {d.max():.3e}  p50 {d.flatten().quantile(0.5):.3e}  p90 {d.flatten().quantile(0.9):.3e}  p99 {d.flatten().quantile(0.99):.3e}")
q("backbone_DA", out["da_output"]["backbone_DA"], g["backbone_DA"])
q("init_box_proposal", out["interm_outputs_for_matching_pre"]["pred_boxes"], g["init_box_proposal"])
same = (out["interm_outputs_for_matching_pre"]["pred_boxes"].float().cpu() - t(g["init_box_proposal"])).abs().amax(-1)[0] < 1e-6
print("selected tokens identical at rank:", int(same.sum()), "/ 900")
q("interm_logits", out["interm_outputs"]["pred_logits"], g["interm_logits"])
q("interm_boxes", out["interm_outputs"]["pred_boxes"], g["interm_boxes"])
q("aux0 logits", out["aux_outputs"][0]["pred_logits"], g["aux_logits"][0])
q("pred_logits", out["pred_logits"], g["pred_logits"])
q("pred_boxes", out["pred_boxes"], g["pred_boxes"])
q("dn_logits", out["dn_meta"]["output_known_lbs_bboxes"]["pred_logits"], g["dn_logits"])
q("proto_source", out["da_output"]["global_proto_DA"]["output_source"], g["proto_source"])
print("total", float(total), float(g["total_loss"]))
for k, v, r in zip(loss_dict.keys(), loss_dict.values(), g["loss_values"]):
    if abs(float(v) - r) > 0.01 * abs(r) + 1e-3: print("  loss", k, float(v), r)
print("tf32 flags:", torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32, torch.get_float32_matmul_precision())
