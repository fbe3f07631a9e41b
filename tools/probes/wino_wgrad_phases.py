"""Phase ablation of csrc/wino_wgrad.hip (development builds with -DWGRAD_ABLATE=<bits>, wrong
results): times the 256 -> 256 layer on the 100x167 level of 4 images with pieces compiled out.
    for a in 1 2 ...; do DATR_HIP_LIB=datr_amd/lib/libdatr_hip_wwabl$a.so python tools/probes/wino_wgrad_phases.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.wino import wino_wgrad  # noqa: E402

dev = torch.device("cuda:0")
w = torch.zeros(256, 256, 3, 3, device=dev)
x = torch.randn(4, 256, 100, 167, device=dev).contiguous(memory_format=torch.channels_last)
dy = torch.randn(4, 256, 100, 167, device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(3):
    wino_wgrad([x], [dy], w)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    wino_wgrad([x], [dy], w)
b.record()
torch.cuda.synchronize()
print(os.environ.get("DATR_HIP_LIB", "product"), f"{a.elapsed_time(b) / 10 * 1e3:.0f} us")
