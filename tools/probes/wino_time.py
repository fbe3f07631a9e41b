"""Times one Winograd/MFMA convolution (CIN -> COUT on the 100x167 level of 4 images); used with
DATR_HIP_LIB=<ablation build> to see what a piece of the kernel costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.domain import wino_conv3x3, wino_filter  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cin, cout = int(os.environ.get("CIN", 256)), int(os.environ.get("COUT", 256))
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.01
b = torch.zeros(cout, device=dev)
x = torch.randn(4, cin, 100, 167, device=dev).contiguous(memory_format=torch.channels_last)
u = wino_filter(w)
for _ in range(5):
    wino_conv3x3([x], u, cout, shift=b, slope=0.2)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    wino_conv3x3([x], u, cout, shift=b, slope=0.2)
e.record()
torch.cuda.synchronize()
print(f"{os.environ.get('DATR_HIP_LIB', 'product')}: {a.elapsed_time(e) / 20 * 1e3:.0f} us")
