"""Feasibility probe: can ONE whole training step (forward + criterion + backward + clip + AdamW) at a fixed shape be
captured into a HIP graph and replayed?  Prints either the replayed step time beside the eager one, or the error that
stops the capture (what would have to change).  Usage: python tools/probes/graph_capture_probe.py [height width]"""
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import tuning  # noqa: E402
from datr_amd.training import build_training, run_steps, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 800)
tuning.enable()
os.environ.setdefault("DATR_MSDA_ADAPTIVE", "0")
state = build_training(device=dev)
batch = synthetic_batch(2, H, W, 10, dev, seed=1)
run_steps(state, [batch] * 6)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_steps(state, [batch] * 10)
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms per step at {H} x {W}")
from datr_amd.engine import _backward_and_step, weighted_total  # noqa: E402
model, criterion, optimizer = state.model, state.criterion, state.optimizer
samples, targets = batch[0], [dict(t) for t in batch[1]]
max_norm = 0.1


def inner_step():
    """The step without what cannot be captured by design: no loss fetch, no reduce_dict."""
    outputs = model(samples, targets)
    loss_dict = criterion(outputs, targets)
    losses = weighted_total(loss_dict, criterion.weight_dict)
    _backward_and_step(model, optimizer, losses, max_norm, None, False, None)
    return losses


stages = {
    "forward": lambda: model(samples, targets),
    "forward+criterion": lambda: weighted_total(criterion(model(samples, targets), targets), criterion.weight_dict),
    "forward+criterion+backward": lambda: (optimizer.zero_grad(), weighted_total(
        criterion(model(samples, targets), targets), criterion.weight_dict).backward()),
    "whole inner step": inner_step,
}
which = os.environ.get("STAGE", "whole inner step")
fn = stages[which]
if hasattr(criterion, "prefetch_num_boxes"):
    criterion.prefetch_num_boxes(targets, dev)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()                                        # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode=os.environ.get("CAPTURE_MODE", "global")):
            fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print(f"{which}: graph replay {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms per step")
except Exception as e:                                   # noqa: BLE001
    print(f"{which}: capture failed:", type(e).__name__, str(e)[:300].replace("\n", " "))
    for fr in traceback.extract_tb(sys.exc_info()[2])[-10:]:
        print("   ", fr.filename.replace("/root/repo/", ""), fr.lineno, fr.name)
