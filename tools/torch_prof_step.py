"""torch.profiler view of one steady-state training step (ATen op names + shapes)."""
import os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import training as bench
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=60); a = ap.parse_args()
class A:
    flat_grads = False
    tuned_gemm = True
dev = torch.device("cuda:0")
tr = bench.Stepper(dev)
samples, targets = bench.synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(3): tr.step(samples, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(samples, targets)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=a.rows, max_name_column_width=40, max_shapes_column_width=70))
