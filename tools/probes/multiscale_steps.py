"""Training steps over the reference's multi-scale resize range (shorter side 480 ... 800 in steps of 32, aspect 5:3, as
`data_aug_scales` of the DA configs give for Cityscapes-shaped images): ms per step by size, three passes, so that
first-visit costs (plan search, library heuristics, allocator growth) show up against the steady state.
Usage: python tools/probes/multiscale_steps.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import tuning  # noqa: E402
from datr_amd.training import build_training, run_steps, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tuning.enable()
state = build_training(device=dev)
sizes = [(h, int(round(h * 1333 / 800))) for h in range(480, 801, 32)]
pool = {s: synthetic_batch(2, s[0], s[1], 10, dev, seed=s[0]) for s in sizes}
run_steps(state, [pool[sizes[-1]]] * 3)
torch.cuda.synchronize()
for rep in range(3):
    row = []
    for s in sizes:
        t0 = time.perf_counter()
        run_steps(state, [pool[s]])
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) * 1e3)
    print(f"pass {rep}: " + "  ".join(f"{s[0]}x{s[1]}: {ms:6.1f}" for s, ms in zip(sizes, row)))
# steady state of a mixed stream
order = [sizes[(7 * i) % len(sizes)] for i in range(33)]
t0 = time.perf_counter()
run_steps(state, [pool[s] for s in order])
torch.cuda.synchronize()
mixed = (time.perf_counter() - t0) / len(order) * 1e3
pix = sum(s[0] * s[1] for s in order) / len(order)
print(f"mixed stream: {mixed:.1f} ms per step at {pix / (800 * 1333):.2f} of the 800 x 1333 pixel count "
      f"({torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak)")
