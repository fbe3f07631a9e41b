"""The stride-1 3x3 layers of the step on csrc/wino.hip, one launch each (forward form), HIP events:
    python tools/probes/wino_layers.py          (DATR_HIP_LIB selects a variant build)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import wino
dev = torch.device("cuda:0")
torch.manual_seed(0)
tot = 0.0
for name, c, h, w, reps in [("layer1 64ch", 64, 200, 334, 3), ("layer2 128ch", 128, 100, 167, 3 + 3), ("layer3 256ch", 256, 50, 84, 5 + 5),
                            ("layer4 512ch", 512, 25, 42, 2 + 2), ("D_img 256>256 l0", 256, 100, 167, 2)]:
    x = torch.randn(4, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
    u = wino.wino_filter(wt)
    sc = torch.rand(c, device=dev); sh = torch.randn(c, device=dev)
    f = lambda: wino.wino_conv3x3([x], u, c, shift=sh, scale=sc, slope=0.0)
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 100
    tot += us * reps
    print(f"{name:20s} {us:7.1f} us  ({2 * 4 * h * w * 9 * c * c / us * 1e-6:6.1f} TF/s direct-equivalent)")
print(f"weighted by launches per step: {tot / 1e3:.2f} ms")
