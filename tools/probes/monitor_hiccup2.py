import os, sys, time, gc
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd.training import Stepper, synthetic_batch
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "nogc":
    gc.disable()
dev = torch.device("cuda:0")
tr = Stepper(dev)
b = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
slow = []
for i in range(60):
    t = time.perf_counter(); tr.step(*b); torch.cuda.synchronize(); dt = time.perf_counter() - t
    if i > 3 and dt > 0.083:
        slow.append((i + 1, round(dt * 1e3, 1)))
print(mode or "default", "slow steps:", slow, "memory reserved GB", round(torch.cuda.memory_reserved() / 2**30, 1),
      "alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"), "device mallocs", torch.cuda.memory_stats().get("num_device_alloc"))
