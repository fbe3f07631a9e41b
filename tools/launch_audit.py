"""Which python lines launch the most kernels in one training step (torch.profiler with stacks)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import training as bench
from torch.profiler import profile, ProfilerActivity


class A:
    flat_grads = False
    tuned_gemm = True
    channels_last = True


dev = torch.device("cuda:0")
tr = bench.Stepper(dev)
samples, targets = bench.synthetic_batch(2, 800, 1333, 10, dev, seed=1)
samples.tensors = samples.tensors.contiguous(memory_format=torch.channels_last)
for _ in range(3):
    tr.step(samples, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(samples, targets)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
by_line = collections.Counter()
by_op = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    nk = len(ev.kernels) if hasattr(ev, "kernels") else 0
    if nk == 0:
        continue
    by_op[ev.name] += nk
    frame = next((s for s in (ev.stack or []) if "/datr_amd/" in s or "/bench.py" in s), None)
    by_line[(frame or "autograd / other").replace(ROOT, "")] += nk
print("kernels by op:")
for k, v in by_op.most_common(25):
    print(f"  {v:5d}  {k}")
print("kernels by source line (forward; backward nodes are 'autograd / other'):")
for k, v in by_line.most_common(45):
    print(f"  {v:5d}  {k}")
