"""Own stride-2 convolution kernels (csrc/conv_tap.hip) against the library at the training step's
shapes (4 images of 1333x800: stage inputs 200x334, 100x167, 50x84; C5 25x42).

    python tools/bench_conv_s2.py
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import _native  # noqa: E402
from datr_amd.strided import _workspace  # noqa: E402

LAYERS = [
    ("layer2.0.conv2", 128, 128, 200, 334, 3),
    ("layer3.0.conv2", 256, 256, 100, 167, 3),
    ("layer4.0.conv2", 512, 512, 50, 84, 3),
    ("input_proj.3", 2048, 256, 25, 42, 3),
]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    N = 4
    lib = _native.lib
    for name, ci, co, H, W, k in LAYERS:
        x = torch.randn(N, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        dy = torch.randn(N, co, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        wt = w.permute(2, 3, 1, 0).contiguous()
        wtt = w.permute(2, 3, 0, 1).contiguous()
        ws = _workspace(x.shape, co, dev)
        st = _native.current_stream_ptr(dev)
        s = dw.stride()
        own_f = timed(lambda: lib.datr_conv3x3s2_forward_nhwc_f32(x.data_ptr(), wt.data_ptr(), 0, 0, 1.0, N, H, W, ci, co,
                                                                y.data_ptr(), ws.data_ptr(), ws.numel(), st))
        own_d = timed(lambda: lib.datr_conv3x3s2_dgrad_nhwc_f32(dy.data_ptr(), wtt.data_ptr(), N, H, W, ci, co,
                                                              dx.data_ptr(), ws.data_ptr(), ws.numel(), st))
        own_w = timed(lambda: lib.datr_conv3x3s2_wgrad_nhwc_f32(x.data_ptr(), dy.data_ptr(), N, H, W, ci, co, dw.data_ptr(),
                                                              s[0], s[1], s[2], s[3], ws.data_ptr(), ws.numel(), st))
        wc = w.contiguous(memory_format=torch.channels_last)
        bwd = torch.ops.aten.convolution_backward
        lib_f = timed(lambda: F.conv2d(x, wc, stride=2, padding=k // 2))
        lib_d = timed(lambda: bwd(dy, x, wc, None, [2, 2], [k // 2] * 2, [1, 1], False, [0, 0], 1, [True, False, False]))
        lib_w = timed(lambda: bwd(dy, x, wc, None, [2, 2], [k // 2] * 2, [1, 1], False, [0, 0], 1, [False, True, False]))
        gf = 2.0 * N * Ho * Wo * ci * co * k * k / 1e9
        print(f"{name:16s} {gf:6.1f} GF | fwd own {own_f:6.0f} us ({gf / own_f * 1e3:5.1f} TF/s) lib {lib_f:6.0f} | "
              f"dgrad own {own_d:6.0f} lib {lib_d:6.0f} | wgrad own {own_w:6.0f} lib {lib_w:6.0f}", flush=True)


if __name__ == "__main__":
    main()
