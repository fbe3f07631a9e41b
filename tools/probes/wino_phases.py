"""Cycle counters of the Winograd kernel's main-loop phases (probe build: -DWINO_PROBE, the `shift`
pointer receives the counters): issue / multiply+transform / DMA wait / barrier, per wave."""
import os
import sys

import torch

os.environ.setdefault("DATR_HIP_LIB", "datr_amd/lib/libdatr_hip_wp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.domain import wino_conv3x3, wino_filter  # noqa: E402

dev = torch.device("cuda:0")
cin, cout = 256, 256
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.01
x = torch.randn(4, cin, 100, 167, device=dev).contiguous(memory_format=torch.channels_last)
u = wino_filter(w)
dbg = torch.zeros(4096, device=dev)
for _ in range(3):
    wino_conv3x3([x], u, cout, shift=dbg, slope=0.2)
torch.cuda.synchronize()
d = dbg[:128].view(8, 4, 4).cpu()
print("per chunk (32 chunks), cycles: [issue, multiply+transform, dma wait, barrier]")
for b in range(8):
    for wv in range(4):
        print(b, wv, [round(float(v) / 32) for v in d[b, wv]])
