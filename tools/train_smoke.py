"""A few dozen optimizer steps of the bench workload on one GPU: the loss must stay finite and go
down on a fixed batch (end-to-end sanity of forward, criterion, backward, clip, AdamW)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import training as bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--height", type=int, default=512)
ap.add_argument("--width", type=int, default=768)
a = ap.parse_args()
args = argparse.Namespace(tuned_gemm=True, channels_last=True, flat_grads=False)
dev = torch.device("cuda:0")
tr = bench.Stepper(dev)
for g in tr.optimizer.param_groups:          # the schedule's lr is tuned for 36 epochs; speed it up
    g["lr"] = g["lr"] * 2
samples, targets = bench.synthetic_batch(2, a.height, a.width, 6, dev, seed=3)
hist = []
for i in range(a.steps):
    loss = tr.step(samples, targets)["loss"]
    if i % 5 == 0 or i == a.steps - 1:
        hist.append(float(loss))
        print(f"step {i:3d} loss {hist[-1]:.4f}", flush=True)
assert all(h == h and abs(h) < 1e6 for h in hist), "non-finite loss"
assert hist[-1] < hist[0], "loss did not decrease on a fixed batch"
print("ok: loss", hist[0], "->", hist[-1])
