"""The reference's strong (photometric) augmentation of the target-domain images on the device.

Mirror of `make_coco_strong_transforms` (/root/reference/datasets/DAcoco.py:348-360) and its
`GaussianBlur` (:330-345): torchvision's `RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8)`,
`RandomGrayscale(p=0.2)`, `RandomApply([GaussianBlur([0.1, 2.0])], p=0.5)` on PIL images.  Here the
image is a uint8 [H, W, 3] DEVICE tensor and the pixel work is csrc/strong_aug.hip, bit-exact with
the Pillow calls torchvision makes for PIL inputs (oracle/pillow_ops.py states and pins them).

torchvision itself is not vendored by the reference and not installed here; the order in which its
transforms draw random numbers is restated from its published source (0.15: RandomApply skips when
`p < torch.rand(1)`; ColorJitter.get_params draws `torch.randperm(4)` then brightness, contrast,
saturation, hue with `torch.empty(1).uniform_`; RandomGrayscale fires when `torch.rand(1) < p`; the
reference's GaussianBlur draws `random.uniform` from Python's generator).  Consecutive per-pixel
transforms are fused: the jitter's four steps and the grayscale become ONE chain handed to
`datr_pixel_ops_u8`.
"""
from __future__ import annotations

import ctypes
import math
import random
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _native

BRIGHTNESS, CONTRAST, SATURATION, HUE, GRAYSCALE = range(5)
PIXEL_OPS_MAX = 8
PixelOp = Tuple[int, float]                     # (code, factor); HUE's factor is the hue shift in [-0.5, 0.5]


class _PixelOp(ctypes.Structure):
    """`datr_pixel_op` of include/datr_hip.h."""
    _fields_ = [("code", ctypes.c_int32), ("alpha", ctypes.c_float), ("shift", ctypes.c_int32)]


def _check_image(image: torch.Tensor, what: str):
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise ValueError("image must be uint8 [H, W, 3]")
    if not image.is_cuda:
        raise RuntimeError(f"{what}: Not implemented on the CPU")


def pixel_ops_on_device(image: torch.Tensor, ops: Sequence[PixelOp]) -> torch.Tensor:
    """Apply a chain of (code, factor) per-pixel operations in one pass (plus one read-only pass per
    CONTRAST step for its mean grey)."""
    _check_image(image, "pixel_ops_on_device")
    if len(ops) > PIXEL_OPS_MAX:
        raise ValueError(f"at most {PIXEL_OPS_MAX} operations per chain")
    if not ops:
        return image
    src = image.contiguous()
    dst = torch.empty_like(src)
    arr = (_PixelOp * len(ops))()
    for k, (code, factor) in enumerate(ops):
        arr[k].code, arr[k].alpha = int(code), float(factor)
        arr[k].shift = (int(factor * 255) & 0xFF) if code == HUE else 0
    sums = (torch.empty(PIXEL_OPS_MAX, dtype=torch.int64, device=image.device)
            if any(c == CONTRAST for c, _ in ops) else None)
    with _native.on_device(image.device):
        rc = _native.lib.datr_pixel_ops_u8(src.data_ptr(), dst.data_ptr(), src.shape[0] * src.shape[1],
                                           ctypes.cast(arr, ctypes.c_void_p), len(ops),
                                           0 if sums is None else sums.data_ptr(),
                                           _native.current_stream_ptr(image.device))
    _native.check(rc, "pixel_ops_u8")
    return dst


def gaussian_box_radius(sigma: float, passes: int = 3) -> float:
    """libImaging BoxBlur.c `_gaussian_blur_radius`: the (fractional) box radius whose `passes`-fold
    box blur has the variance of a Gaussian of std `sigma`; float32 variables as in C."""
    f = np.float32
    sigma2 = f(f(sigma) * f(sigma) / passes)
    L = f(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(math.floor((float(L) - 1.0) / 2.0))
    a = f(f(2 * l + 1) * f(f(l * f(l + 1)) - f(3 * sigma2)))
    a = f(a / f(6 * f(sigma2 - f(f(l + 1) * f(l + 1)))))
    return float(f(l + a))


def box_weights(float_radius: float) -> Tuple[int, int, int]:
    """(radius, ww, fw) of libImaging BoxBlur.c `ImagingLineBoxBlur*`: 8.24 fixed-point weights of
    the 2 radius + 1 inner taps and of the two far taps."""
    radius = int(float_radius)
    ww = int(np.uint32(np.float32(1 << 24) / np.float32(np.float32(float_radius) * 2 + 1)))
    fw = (((1 << 24) - (radius * 2 + 1) * ww) // 2) & 0xFFFFFFFF
    return radius, ww, fw


def gaussian_blur_on_device(image: torch.Tensor, sigma: float, passes: int = 3) -> torch.Tensor:
    """`image.filter(ImageFilter.GaussianBlur(radius=sigma))`."""
    _check_image(image, "gaussian_blur_on_device")
    fr = gaussian_box_radius(sigma, passes)
    if fr == 0:
        return image
    radius, ww, fw = box_weights(fr)
    src = image.contiguous()
    dst = torch.empty_like(src)
    with _native.on_device(image.device):
        rc = _native.lib.datr_box_blur_u8(src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], radius, ww, fw,
                                          passes, _native.current_stream_ptr(image.device))
    _native.check(rc, "box_blur_u8")
    return dst


# ---- the transform objects (torchvision's names and constructor arguments) ---------------------------------

def _jitter_range(value, center=1.0, bound=(0.0, float("inf")), clip_first_on_zero=True):
    """torchvision ColorJitter._check_input."""
    if isinstance(value, (int, float)):
        if value < 0:
            raise ValueError("jitter strength must be non-negative")
        lo, hi = center - float(value), center + float(value)
        if clip_first_on_zero:
            lo = max(lo, 0.0)
    else:
        lo, hi = float(value[0]), float(value[1])
    if not bound[0] <= lo <= hi <= bound[1]:
        raise ValueError(f"jitter values should be between {bound}")
    return None if lo == hi == center else (lo, hi)


class ColorJitter:
    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        self.brightness = _jitter_range(brightness)
        self.contrast = _jitter_range(contrast)
        self.saturation = _jitter_range(saturation)
        self.hue = _jitter_range(hue, center=0, bound=(-0.5, 0.5), clip_first_on_zero=False)

    def draw(self) -> List[PixelOp]:
        order = torch.randperm(4).tolist()
        factors = [None if r is None else float(torch.empty(1).uniform_(r[0], r[1]))
                   for r in (self.brightness, self.contrast, self.saturation, self.hue)]
        return [(k, factors[k]) for k in order if factors[k] is not None]

    def __call__(self, image):
        return pixel_ops_on_device(image, self.draw())


class RandomGrayscale:
    def __init__(self, p=0.1):
        self.p = p

    def draw(self) -> List[PixelOp]:
        return [(GRAYSCALE, 0.0)] if torch.rand(1) < self.p else []

    def __call__(self, image):
        return pixel_ops_on_device(image, self.draw())


class GaussianBlur:
    """The reference's own class (DAcoco.py:330-345), sigma from Python's generator."""

    def __init__(self, sigma=(0.1, 2.0)):
        self.sigma = sigma

    def __call__(self, image):
        return gaussian_blur_on_device(image, random.uniform(self.sigma[0], self.sigma[1]))


class RandomApply:
    def __init__(self, transforms, p=0.5):
        self.transforms, self.p = list(transforms), p

    def draw(self):
        """The pixel-op chain of this draw (empty when skipped); only when every member is per-pixel."""
        if self.p < torch.rand(1):
            return []
        ops: List[PixelOp] = []
        for t in self.transforms:
            ops += t.draw()
        return ops

    @property
    def per_pixel(self):
        return all(hasattr(t, "draw") for t in self.transforms)

    def __call__(self, image):
        if self.per_pixel:
            return pixel_ops_on_device(image, self.draw())
        if self.p < torch.rand(1):
            return image
        for t in self.transforms:
            image = t(image)
        return image


class Compose:
    """torchvision's Compose, with runs of per-pixel transforms fused into one chain."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, image):
        pending: List[PixelOp] = []
        for t in self.transforms:
            if hasattr(t, "draw") and getattr(t, "per_pixel", True):
                pending += t.draw()
                continue
            image = self._flush(image, pending)
            pending = []
            image = t(image)
        return self._flush(image, pending)

    @staticmethod
    def _flush(image, ops):
        while ops:
            image = pixel_ops_on_device(image, ops[:PIXEL_OPS_MAX])
            ops = ops[PIXEL_OPS_MAX:]
        return image


def make_strong_transforms(image_set: str = "train"):
    """`make_coco_strong_transforms` (DAcoco.py:348-360)."""
    if image_set == "train":
        return Compose([RandomApply([ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8), RandomGrayscale(p=0.2),
                        RandomApply([GaussianBlur([0.1, 2.0])], p=0.5)])
    if image_set == "val":
        return None
    raise ValueError(f"unknown {image_set}")


__all__ = ["BRIGHTNESS", "CONTRAST", "SATURATION", "HUE", "GRAYSCALE", "pixel_ops_on_device",
           "gaussian_blur_on_device", "gaussian_box_radius", "box_weights", "ColorJitter", "RandomGrayscale",
           "GaussianBlur", "RandomApply", "Compose", "make_strong_transforms"]
