import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import build_model
from datr_amd.training import synthetic_batch
dev = torch.device("cuda:0")
args, model, criterion, _ = build_model("cuda:0")
criterion.to(dev)
model.backbone.to(memory_format=torch.channels_last)
model.train(); criterion.train()
samples, targets = synthetic_batch(1, 256, 320, 3, dev, seed=1)
for scale in (65536.0, 1024.0, 1.0):
    model.zero_grad(set_to_none=True)
    with torch.autocast(device_type="cuda", enabled=True):
        out = model(samples, list(targets))
        ld = criterion(out, list(targets))
        wd = criterion.weight_dict
        loss = sum(ld[k] * wd[k] for k in ld if k in wd)
    bad_l = [k for k, v in ld.items() if not torch.isfinite(v).all()]
    (loss * scale).backward()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(f"scale {scale}: loss {float(loss):.4f} dtype {loss.dtype} non-finite losses {bad_l[:5]} non-finite grads {len(bad)} e.g. {bad[:6]}")
