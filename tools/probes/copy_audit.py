import os, sys, argparse, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import Stepper, synthetic_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tr = Stepper(dev)
samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(4):
    tr.step(samples, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(samples, targets)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::masked_fill", "aten::masked_fill_", "aten::copy_", "aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::div", "aten::cat", "aten::fill_"):
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
for r in rows[:40]:
    print(f"{r[0] / 1e3:7.3f} ms  {r[1]:4d}x  {r[2]:12s} {r[3]}")
