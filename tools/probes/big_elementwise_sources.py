"""Source lines of the large copies / clones / fills / cats / adds of a training step (forward and the
Python-side of backward nodes): a TorchDispatchMode that records the repo frame of every such op on a
tensor of >= 1 M elements.   python tools/probes/big_elementwise_sources.py
"""
import os
import sys
import traceback
from collections import defaultdict

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import Stepper, synthetic_batch  # noqa: E402

WATCH = ("clone", "copy_", "fill_", "zero_", "cat", "add", "add_", "mul", "zeros", "zeros_like", "contiguous",
         "_to_copy", "masked_fill", "where", "sum")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda and t.numel() >= (1 << 20):
                fr = [f for f in traceback.extract_stack() if "/datr_amd/" in f.filename][-2:]
                key = (name, " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))
                self.agg[key][0] += 1
                self.agg[key][1] += t.numel() * 4
        return out


def main():
    dev = torch.device("cuda:0")
    tr = Stepper(dev, tuned_gemm=True, channels_last=True)
    samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
    for _ in range(2):
        tr.step(samples, targets)
    spy = Spy()
    with spy:
        tr.step(samples, targets)
    torch.cuda.synchronize()
    for (n, where), (c, b) in sorted(spy.agg.items(), key=lambda kv: -kv[1][1])[:50]:
        print(f"{b / 1e6:9.1f} MB {c:3d} x {n:12s} {where}")


if __name__ == "__main__":
    main()
