"""1x1 convolutions of the ResNet-50 bottlenecks at the bench resolution: MIOpen (NHWC, shipped
find-db) vs the same product as a hipBLASLt GEMM on the [N*H*W, Cin] view.  fwd / dgrad / wgrad."""
import os, sys, json
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import tuning
tuning.enable()
dev = torch.device("cuda:0")

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

cases = [(200, 334, 64, 64), (200, 334, 64, 256), (200, 334, 256, 64), (200, 334, 256, 128),
         (100, 167, 128, 512), (100, 167, 512, 128), (100, 167, 512, 256), (50, 84, 256, 1024),
         (50, 84, 1024, 256), (50, 84, 1024, 512), (25, 42, 512, 2048), (25, 42, 2048, 512)]
tot = {"conv": 0.0, "mm": 0.0}
for H, W, ci, co in cases:
    x = torch.randn(4, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(4, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    x2 = x.permute(0, 2, 3, 1).reshape(-1, ci)
    w2 = w.view(co, ci)
    dy2 = dy.permute(0, 2, 3, 1).reshape(-1, co)
    y = F.conv2d(x, w)
    err = (y.permute(0, 2, 3, 1).reshape(-1, co) - x2 @ w2.t()).abs().max().item()
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    def conv_bwd():
        yy = F.conv2d(xr, wr)
        return torch.autograd.grad(yy, (xr, wr), dy)
    r = {"shape": [H, W, ci, co], "err": round(err, 6),
         "conv_fwd": t(lambda: F.conv2d(x, w)), "mm_fwd": t(lambda: x2 @ w2.t()),
         "conv_fwd+bwd": t(conv_bwd),
         "mm_dgrad": t(lambda: dy2 @ w2), "mm_wgrad": t(lambda: dy2.t() @ x2)}
    r = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}
    r["conv_bwd"] = round(r["conv_fwd+bwd"] - r["conv_fwd"], 1)
    r["mm_bwd"] = round(r["mm_dgrad"] + r["mm_wgrad"], 1)
    print(json.dumps(r), flush=True)
