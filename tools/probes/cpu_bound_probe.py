"""Is the step CPU-bound?  Host time to ENQUEUE 20 steps vs time until the GPU has finished them."""
import os, sys, time, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import Stepper, synthetic_batch
dev = torch.device("cuda:0")
tr = Stepper(dev)
samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(6):
    tr.step(samples, targets)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(samples, targets)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 20:.1f} ms/step, complete {1e3 * (t2 - t0) / 20:.1f} ms/step, "
      f"GPU backlog at the end {1e3 * (t2 - t1):.1f} ms")
