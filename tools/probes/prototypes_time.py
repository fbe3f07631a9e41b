"""Time of the class-prototype kernel (csrc/prototypes.hip) against the number of rows.
Usage: python tools/probes/prototypes_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.domain import _ClassPrototypes  # noqa: E402

dev = torch.device("cuda:0")
for R in (550, 2200, 8800):
    feats = torch.randn(R, 256, device=dev)
    labels = torch.randint(0, 9, (R,), device=dev)
    gp, am = torch.randn(9, 256, device=dev), torch.rand(9, device=dev) * 10
    for _ in range(5):
        _ClassPrototypes.apply(feats, labels, gp, am, 9)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        _ClassPrototypes.apply(feats, labels, gp, am, 9)
    b.record()
    torch.cuda.synchronize()
    print(f"R = {R}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call (forward, incl. output allocation)")
