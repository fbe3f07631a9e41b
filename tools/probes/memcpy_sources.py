"""Where do the device-to-device copies and the largest element-wise kernels of a training step come from?
torch.profiler with Python stacks over two steps; prints the copy / add events >= 5 us with their source lines.
    python tools/probes/memcpy_sources.py
"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import Stepper, synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tr = Stepper(dev, tuned_gemm=True, channels_last=True)
    samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
    for _ in range(3):
        tr.step(samples, targets)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.step(samples, targets)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_stack_n=8):
        n = e.key
        if not (n.startswith("aten::copy_") or n.startswith("aten::clone") or n == "aten::add" or n == "aten::add_"
                or n.startswith("aten::where") or n.startswith("aten::fill_") or n.startswith("aten::zero_")
                or n.startswith("aten::cat") or n.startswith("aten::sum") or n.startswith("aten::mul")):
            continue
        t = getattr(e, "device_time_total", None)
        if t is None:
            t = e.cuda_time_total
        if t < 15:
            continue
        stack = [f for f in (e.stack or []) if "repo/" in f and "memcpy_sources" not in f][:3]
        rows.append((t, e.count, n, stack))
    for t, c, n, st in sorted(rows, key=lambda r: -r[0])[:45]:
        print(f"{t:8.1f} us {c:3d} x {n:14s} {' <- '.join(x.split('repo/')[-1].strip() for x in st)}")


if __name__ == "__main__":
    main()
