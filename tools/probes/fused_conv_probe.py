"""Frozen stem / layer1: conv + folded frozen-BN + ReLU as ONE MIOpen fused call vs conv + affine kernel."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import tuning
from datr_amd.fused import frozen_bn_act
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
for (ci, co, k, H, W, res) in ((64, 64, 1, 200, 334, False), (64, 64, 3, 200, 334, False), (64, 256, 1, 200, 334, True),
                               (256, 64, 1, 200, 334, False)):
    x = torch.randn(4, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    scale = torch.rand(co, device=dev) + 0.5
    shift = torch.randn(co, device=dev)
    z = torch.randn(4, co, H, W, device=dev).contiguous(memory_format=torch.channels_last) if res else None
    wf = (w * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    with torch.no_grad():
        ref = frozen_bn_act(F.conv2d(x, w, None, padding=pad), scale, shift, residual=z, relu=True)
        if res:
            fused = lambda: torch.ops.aten.miopen_convolution_add_relu(x, wf, z, 1.0, shift, [1, 1], [pad, pad], [1, 1], 1)
        else:
            fused = lambda: torch.ops.aten.miopen_convolution_relu(x, wf, shift, [1, 1], [pad, pad], [1, 1], 1)
        out = fused()
        err = (out - ref).abs().max().item()
        t_ref = t(lambda: frozen_bn_act(F.conv2d(x, w, None, padding=pad), scale, shift, residual=z, relu=True))
        t_conv = t(lambda: F.conv2d(x, w, None, padding=pad))
        t_fused = t(fused)
    print((ci, co, k, res), "err", round(err, 6), "conv", round(t_conv, 1), "conv+affine", round(t_ref, 1), "fused", round(t_fused, 1))
