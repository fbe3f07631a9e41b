import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from helpers import build_model, force_reference_selection, load_npz, run_training_step, t
dev = torch.device("cuda:0")
g = load_npz("model_step.npz")
for nhwc in (False, True):
    _, model, criterion, _ = build_model("cuda:0")
    if nhwc: model.backbone.to(memory_format=torch.channels_last)
    force_reference_selection(model, g, dev)
    with torch.autocast("cuda", dtype=torch.float16):
        out, loss_dict, indices_list, total = run_training_step(model, criterion, dev, g, channels_last=nhwc)
    d = lambda a, b: float((a.detach().float().cpu() - t(b)).abs().max())
    print("nhwc", nhwc, "logits", d(out["pred_logits"], g["pred_logits"]), "boxes", d(out["pred_boxes"], g["pred_boxes"]),
          "interm", d(out["interm_outputs"]["pred_logits"], g["interm_logits"]), "DA", d(out["da_output"]["backbone_DA"], g["backbone_DA"]),
          "total", float(total), float(g["total_loss"]), out["pred_logits"].dtype)
    import numpy as np
    mine = np.stack([np.stack([np.stack([s.cpu().numpy(), tt.cpu().numpy()]) for s, tt in call]) for call in indices_list])
    print("indices equal:", float((mine == g["indices"]).mean()))
    lv = torch.tensor([float(v.detach()) for v in loss_dict.values()], dtype=torch.float64)
    rel = ((lv - t(g["loss_values"])).abs() / (t(g["loss_values"]).abs() + 1e-3))
    print("loss rel max", float(rel.max()), "median", float(rel.median()))
