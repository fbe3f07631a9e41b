"""What a dependent one-workgroup kernel costs inside the training step: N extra `fill_` launches on a
one-element tensor after the transformer's forward, N = 0 / 200 / 400, step time each (bench shape).
rocprofv3 reports 4.4-5 us for every such kernel of the step; this measures what they add to the wall clock.
Usage: python tools/probes/tiny_kernel_cost.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import build_training, run_steps, synthetic_batch  # noqa: E402
from datr_amd import tuning  # noqa: E402

dev = torch.device("cuda:0")
tuning.enable()
state = build_training(device=dev)
pool = [synthetic_batch(2, 800, 1333, 10, dev, seed=1 + 1000 * i) for i in range(4)]
extra = {"n": 0}
scratch = torch.zeros(1, device=dev)


def hook(module, args, output):
    for _ in range(extra["n"]):
        scratch.fill_(1.0)
    return output


model = state["model"] if isinstance(state, dict) else state.model
model.transformer.register_forward_hook(hook)
run_steps(state, [pool[i % 4] for i in range(8)])
for n in (0, 200, 400, 0, 200, 400):
    extra["n"] = n
    run_steps(state, [pool[i % 4] for i in range(3)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 20
    run_steps(state, [pool[i % 4] for i in range(k)])
    torch.cuda.synchronize()
    print(f"extra launches {n:4d}: {(time.perf_counter() - t0) / k * 1e3:.2f} ms/step")
