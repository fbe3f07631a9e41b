"""Every GEMM-type ATen op of one training step with input shapes, call count and device time
(torch.profiler): which library GEMMs the step still spends its time in.
    python tools/probes/gemm_calls.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import training as bench  # noqa: E402

dev = torch.device("cuda:0")
tr = bench.Stepper(dev)
samples, targets = bench.synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(3):
    tr.step(samples, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(samples, targets)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    t = getattr(e, "self_device_time_total", 0) or 0
    if t <= 0 or e.name not in ("aten::mm", "aten::addmm", "aten::bmm", "aten::_addmm_activation", "aten::addmm_"):
        continue
    p = e.cpu_parent
    node = ""
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function"):
            node = p.name.split(": ")[-1]
            break
        p = p.cpu_parent
    k = (e.name, str(e.input_shapes), node)
    agg[k][0] += t
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"library GEMM ops: {tot / 1e3:.2f} ms in {sum(v[1] for v in agg.values())} calls")
for (n, s, node), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{t:9.1f} us {c:3d} x {t / c:7.1f}  {n:24s} {s:60s} {node}")
