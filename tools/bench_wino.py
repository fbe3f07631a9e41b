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


def backbone_layers():
    """conv2 + frozen BN + ReLU of the ResNet-50 bottlenecks at 4 x 1333x800 (stride-1 blocks):
    own Winograd/MFMA launch vs library convolution + fused frozen-BN pass, forward and fwd+bwd."""
    from datr_amd import wino
    from datr_amd.fused import frozen_bn_act
    wino.OWN_BACKBONE_3X3_MAX_CH = 1 << 20          # time every width, whatever the product routes
    dev = torch.device("cuda:0")
    for c, h, w_, n_blocks, train in ((64, 200, 334, 3, False), (128, 100, 167, 3, True), (256, 50, 84, 5, True),
                                      (512, 25, 42, 2, True)):
        wt = (torch.randn(c, c, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
        wt.requires_grad_(train)
        scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
        x = torch.randn(4, c, h, w_, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(train)
        go = torch.randn(4, c, h, w_, device=dev).contiguous(memory_format=torch.channels_last)
        own = lambda: wino.conv3x3_bn_relu(x, wt, scale, shift)
        lib = lambda: frozen_bn_act(torch.nn.functional.conv2d(x, wt, padding=1), scale, shift, relu=True)
        row = f"backbone conv2 {c}ch @{h}x{w_} (x{n_blocks}/step):"
        for name, fn in (("own", own), ("library", lib)):
            with torch.no_grad():
                t_f = timed(fn, 10)
            t_fb = timed(lambda: torch.autograd.grad(fn(), (x, wt), go), 10) if train else float("nan")
            row += f"  {name} fwd {t_f * 1e3:.0f} us, fwd+bwd {t_fb * 1e3:.0f} us;"
        print(row)


if __name__ == "__main__" and os.environ.get("BENCH_BACKBONE", "1") != "0":
    backbone_layers()
