"""Run-to-run reproducibility of the NHWC training step at the golden size (the configuration of
tests/test_model_gpu.py::test_nhwc_step_with_reference_selection_matches_elementwise): N fresh
models, same inputs; prints the largest gradient differences against the first run."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from helpers import build_model, force_reference_selection, load_npz, run_training_step  # noqa: E402

dev = torch.device("cuda:0")
g = load_npz("model_step.npz")
ref = None
for it in range(int(os.environ.get("RUNS", 8))):
    _, model, criterion, _ = build_model("cuda:0")
    model.backbone.to(memory_format=torch.channels_last)
    force_reference_selection(model, g, dev)
    out, loss_dict, idx, total = run_training_step(model, criterion, dev, g, channels_last=True)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    if ref is None:
        ref = grads
        continue
    worst = sorted(((float((grads[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-12)), k) for k in ref), reverse=True)[:4]
    print(it, "total", float(total), " ".join(f"{k}:{v:.2e}" for v, k in worst))
