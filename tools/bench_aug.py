"""Time the device input-pipeline kernels on a Cityscapes-sized frame (1024 x 2048 uint8 RGB):
flip + resize (csrc/resize.hip), the fused ColorJitter + grayscale chain and the Gaussian blur
(csrc/strong_aug.hip), normalise + pad (csrc/preprocess.hip), with their algorithmic bytes.

    python tools/bench_aug.py [--iters 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import strong_aug as S  # noqa: E402
from datr_amd.input_pipeline import collate_uint8_on_device, resize_uint8_on_device  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (1024, 2048, 3), generator=g, dtype=torch.uint8).to("cuda:0")
    nb = img.numel()
    out = resize_uint8_on_device(img, (666, 1332))
    jitter = [(S.BRIGHTNESS, 1.2), (S.HUE, 0.05), (S.CONTRAST, 0.8), (S.SATURATION, 1.3)]
    cases = {
        "flip+resize 1024x2048 -> 666x1332": (lambda: resize_uint8_on_device(img, (666, 1332), flip=True), nb + out.numel()),
        "jitter chain (4 ops, contrast 3rd)": (lambda: S.pixel_ops_on_device(img, jitter), 3 * nb),
        "jitter chain + grayscale": (lambda: S.pixel_ops_on_device(img, jitter + [(S.GRAYSCALE, 0.0)]), 3 * nb),
        "brightness only": (lambda: S.pixel_ops_on_device(img, jitter[:1]), 2 * nb),
        "hue only": (lambda: S.pixel_ops_on_device(img, jitter[1:2]), 2 * nb),
        "gaussian blur sigma 2.0": (lambda: S.gaussian_blur_on_device(img, 2.0), 2 * nb),
        "gaussian blur sigma 0.5": (lambda: S.gaussian_blur_on_device(img, 0.5), 2 * nb),
        "normalise + pad 666x1332": (lambda: collate_uint8_on_device([out]), out.numel() * 5 + out.numel() // 3),
    }
    for name, (fn, nbytes) in cases.items():
        us = timed(fn, args.iters)
        print(json.dumps({"case": name, "us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 2),
                          "GB_per_s": round(nbytes / us / 1e3, 1)}))


if __name__ == "__main__":
    main()
