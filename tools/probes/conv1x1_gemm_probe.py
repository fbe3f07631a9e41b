"""ResNet-50 1x1 convolutions at 4 x 1333x800 in NHWC: MIOpen (F.conv2d and its backward) against the
same arithmetic as plain GEMMs on the [pixels, channels] view (hipBLASLt through torch.mm, TunableOp
picking the kernel).  fwd / dgrad / wgrad in us."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import tuning  # noqa: E402

tuning.enable(tune=os.environ.get("TUNE", "1") == "1")
dev = torch.device("cuda:0")


def timed(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


shapes = [(200, 334, 64, 64, 1), (200, 334, 64, 256, 4), (200, 334, 256, 64, 2), (200, 334, 256, 128, 1),
          (100, 167, 128, 512, 4), (100, 167, 512, 128, 3), (100, 167, 512, 256, 2), (50, 84, 256, 1024, 6),
          (50, 84, 1024, 256, 6), (50, 84, 1024, 512, 1), (25, 42, 512, 2048, 3), (25, 42, 2048, 512, 2),
          (25, 42, 2048, 256, 1)]
tot = {"miopen": 0.0, "gemm": 0.0}
conv_bwd = torch.ops.aten.convolution_backward
for h, w, ci, co, cnt in shapes:
    x = torch.randn(4, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(4, co, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    x2, w2, dy2 = x.permute(0, 2, 3, 1).reshape(-1, ci), wt.reshape(co, ci), dy.permute(0, 2, 3, 1).reshape(-1, co)
    assert x2.data_ptr() == x.data_ptr() and dy2.data_ptr() == dy.data_ptr()
    m = [timed(lambda: torch.nn.functional.conv2d(x, wt)),
         timed(lambda: conv_bwd(dy, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])),
         timed(lambda: conv_bwd(dy, x, wt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))]
    g = [timed(lambda: x2.mm(w2.t())), timed(lambda: dy2.mm(w2)), timed(lambda: dy2.t().mm(x2))]
    fl = 2 * 4 * h * w * ci * co / 1e6
    print(f"{h}x{w} {ci:4d}->{co:4d} x{cnt}: MIOpen fwd/dgrad/wgrad {m[0]:6.0f} {m[1]:6.0f} {m[2]:6.0f} us | "
          f"GEMM {g[0]:6.0f} {g[1]:6.0f} {g[2]:6.0f} us | GEMM TF/s {fl / g[0]:5.0f} {fl / g[1]:5.0f} {fl / g[2]:5.0f}")
    train = 0 if h == 200 and ci != 256 or (h == 200 and co == 64) else 1
    tot["miopen"] += cnt * (m[0] + train * (m[1] + m[2]))
    tot["gemm"] += cnt * (g[0] + train * (g[1] + g[2]))
print(f"per step (fwd for all, + dgrad + wgrad for the trainable layers): MIOpen {tot['miopen'] / 1e3:.2f} ms, "
      f"GEMM {tot['gemm'] / 1e3:.2f} ms")
