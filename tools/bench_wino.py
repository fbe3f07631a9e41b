"""Times the image-level discriminator (FCDiscriminator_img on the 4-level pyramid of 4 images at
1333x800) forward and backward: own Winograd/MFMA path (csrc/wino.hip) vs the library convolutions
under autograd.  Usage: python tools/bench_wino.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import domain, tuning  # noqa: E402


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
    if hasattr(tuning, "enable"):
        tuning.enable()
    torch.manual_seed(0)
    d = domain.FCDiscriminator_img(256).to(dev)
    sizes = [(100, 167), (50, 84), (25, 42), (13, 21)]
    xs = [torch.randn(4, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for h, w in sizes]
    gos = [torch.randn(4, 1, h, w, device=dev) for h, w in sizes]
    flops = sum(2 * 4 * h * w * 9 * (256 * 256 + 256 * 128 + 128 * 128 + 128) for h, w in sizes)
    for own in (True, False):
        domain.OWN_D_IMG = own
        with torch.no_grad():
            t_f = timed(lambda: d.reversed_pyramid(xs))

        def step():
            outs = d.reversed_pyramid(xs)
            torch.autograd.grad(outs, list(d.parameters()) + xs, gos)
        t_fb = timed(step)
        print(f"{'own winograd/mfma' if own else 'library convs    '}: fwd {t_f:.3f} ms ({flops / t_f / 1e9:.0f} TF/s "
              f"direct-equivalent)  fwd+bwd {t_fb:.3f} ms ({3 * flops / t_fb / 1e9:.0f} TF/s)")
    # single layers, level 0 only
    from datr_amd.domain import wino_conv3x3, wino_filter
    for cin, cout in ((256, 256), (256, 128), (128, 128)):
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.01
        b = torch.zeros(cout, device=dev)
        x = torch.randn(4, cin, 100, 167, device=dev).contiguous(memory_format=torch.channels_last)
        u = wino_filter(w)
        t_o = timed(lambda: wino_conv3x3([x], u, cout, shift=b, slope=0.2), 20)
        t_l = timed(lambda: torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, w, b, padding=1), 0.2), 20)
        fl = 2 * 4 * 100 * 167 * 9 * cin * cout
        print(f"layer {cin}->{cout} @100x167x4: own {t_o * 1e3:.0f} us ({fl / t_o / 1e9:.0f} TF/s eq.)  "
              f"library conv+bias+lrelu {t_l * 1e3:.0f} us ({fl / t_l / 1e9:.0f} TF/s)")


if __name__ == "__main__":
    main()
