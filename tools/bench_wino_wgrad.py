"""Times the Winograd-domain weight gradient (csrc/wino_wgrad.hip) against the library's
weight-gradient convolution on the shapes of the 1333x800 step: the discriminator's three layers
over the 4-level pyramid of 4 images, and conv2 of the trainable ResNet stages.
Usage: python tools/bench_wino_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import tuning  # noqa: E402
from datr_amd.wino import wino_wgrad  # noqa: E402


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    tuning.enable()
    torch.manual_seed(0)
    pyramid = [(100, 167), (50, 84), (25, 42), (13, 21)]
    cases = [("D_img 256->256 pyramid", 256, 256, pyramid), ("D_img 256->128 pyramid", 256, 128, pyramid),
             ("D_img 128->128 pyramid", 128, 128, pyramid), ("256->256 @100x167 only", 256, 256, pyramid[:1]),
             ("layer2 128->128 @100x167", 128, 128, [(100, 167)]), ("layer3 256->256 @50x84", 256, 256, [(50, 84)]),
             ("layer4 512->512 @25x42", 512, 512, [(25, 42)])]
    conv_bwd = torch.ops.aten.convolution_backward
    for name, cin, cout, sizes in cases:
        w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.01).contiguous(memory_format=torch.channels_last)
        xs = [torch.randn(4, cin, h, ww, device=dev).contiguous(memory_format=torch.channels_last) for h, ww in sizes]
        dys = [torch.randn(4, cout, h, ww, device=dev).contiguous(memory_format=torch.channels_last) for h, ww in sizes]

        def lib():
            dw = None
            for x, dy in zip(xs, dys):
                _, g, _ = conv_bwd(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
                dw = g if dw is None else dw.add_(g)
            return dw
        t_o = timed(lambda: wino_wgrad(xs, dys, w))
        t_l = timed(lib)
        fl = sum(2 * 4 * h * ww * 9 * cin * cout for h, ww in sizes)
        err = (wino_wgrad(xs, dys, w) - lib()).abs().max().item() / lib().abs().max().item()
        print(f"{name:28s} own {t_o * 1e3:7.0f} us ({fl / t_o / 1e9:5.0f} TF/s direct-eq.)   library {t_l * 1e3:7.0f} us "
              f"({fl / t_l / 1e9:5.0f} TF/s)   rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
