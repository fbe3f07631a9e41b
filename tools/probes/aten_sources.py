"""Where the ATen element-wise / copy / reduction kernels of one training step come from: self device time
of every aten:: op grouped by the nearest datr_amd/ source line (forward) or the autograd node that ran it
(backward), own C-ABI kernels and GEMMs excluded.
    python tools/probes/aten_sources.py [--rows 90]"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import training as bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=90)
a = ap.parse_args()
dev = torch.device("cuda:0")
tr = bench.Stepper(dev)
samples, targets = bench.synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(3):
    tr.step(samples, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(samples, targets)
    torch.cuda.synchronize()

SKIP = ("aten::mm", "aten::addmm", "aten::bmm", "aten::_addmm_activation", "aten::convolution", "aten::linear", "aten::matmul")
agg = collections.defaultdict(lambda: [0.0, 0])
total = 0.0
for e in prof.events():
    t = getattr(e, "self_device_time_total", 0) or 0
    if t <= 0 or not e.name.startswith("aten::") or e.name.startswith(SKIP):
        continue
    where = None
    for fr in (e.stack or []):
        if "datr_amd/" in fr:
            where = fr.split("datr_amd/")[-1].strip()
            break
    if where is None:
        p = e.cpu_parent
        while p is not None:
            if p.name.startswith("autograd::engine::evaluate_function"):
                where = "bwd:" + p.name.split(": ")[-1]
                break
            for fr in (p.stack or []):
                if "datr_amd/" in fr:
                    where = fr.split("datr_amd/")[-1].strip()
                    break
            if where:
                break
            p = p.cpu_parent
    if where is None:
        where = "?"
    shapes = str(e.input_shapes)[:60] if e.input_shapes else ""
    k = (where[:70], e.name, shapes)
    agg[k][0] += t
    agg[k][1] += 1
    total += t
print(f"ATen (non-GEMM) self device time: {total / 1e3:.2f} ms in {sum(v[1] for v in agg.values())} ops")
by_where = collections.defaultdict(lambda: [0.0, 0])
for (w, n, s), (t, c) in agg.items():
    by_where[w][0] += t
    by_where[w][1] += c
print("---- by source")
for w, (t, c) in sorted(by_where.items(), key=lambda kv: -kv[1][0])[:a.rows]:
    print(f"{t:9.1f} us {c:4d} ops  {w}")
print("---- by (source, op, shapes)")
for (w, n, s), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.rows]:
    print(f"{t:9.1f} us {c:4d}  {n:28s} {w}  {s}")
