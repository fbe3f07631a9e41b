"""Device-side tail of the input pipeline (first piece).

The reference's data loader produces, per image, a normalised float CHW tensor on the host
(`T.ToTensor` + `T.Normalize`, /root/reference/datasets/da_transforms.py:250-276) and
`collate_fn_da` -> `nested_tensor_from_tensor_list` (/root/reference/util/misc.py:291-300,
:387-409) pads them into one batch with a bool mask.  `collate_uint8_on_device` takes the
uint8 HWC images instead (what the decoder / resize / flip stages hand over), moves a quarter
of the bytes across PCIe, and builds the padded batch -- directly in the backbone's NHWC layout
if asked -- and the mask in one kernel per image (csrc/preprocess.hip).  Same values bit for bit.
`resize_uint8_on_device` is the RandomResize / RandomHorizontalFlip stage in front of it: Pillow's
8-bit bilinear resampler restated (weights on the host, two small kernels), bit-exact.  The
transform objects that drive it live in datr_amd/transforms.py, the strong-augmentation kernels
(ColorJitter, grayscale, blur) in datr_amd/strong_aug.py; `collate_fn_da_on_device` is the
reference's `collate_fn_da` for their uint8 device output.
"""
from __future__ import annotations

import ctypes
import math
from typing import Sequence

import numpy as np

import torch

from . import _native
from .nested import NestedTensor

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # datasets/DAcoco.py normalisation constants
IMAGENET_STD = (0.229, 0.224, 0.225)


def collate_uint8_on_device(images: Sequence[torch.Tensor], device=None, mean=IMAGENET_MEAN,
                            std=IMAGENET_STD, channels_last: bool = True) -> NestedTensor:
    """images: uint8 [H_i, W_i, 3] tensors (host or device) -> NestedTensor with
    tensors [B, 3, Hmax, Wmax] float32 (memory format channels_last if asked), mask [B, Hmax, Wmax]
    bool (True = padding) and the host-known `padded` flag."""
    if not images:
        raise ValueError("empty batch")
    device = torch.device(device) if device is not None else images[0].device
    if device.type != "cuda":
        raise RuntimeError("collate_uint8_on_device: Not implemented on the CPU")
    for im in images:
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
            raise ValueError("images must be uint8 [H, W, 3]")
    B = len(images)
    Hp = max(int(im.shape[0]) for im in images)
    Wp = max(int(im.shape[1]) for im in images)
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    batch = torch.empty((B, 3, Hp, Wp), dtype=torch.float32, device=device, memory_format=fmt)
    mask = torch.empty((B, Hp, Wp), dtype=torch.bool, device=device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    slot = 3 * Hp * Wp * 4
    with _native.on_device(device):
        stream = _native.current_stream_ptr(device)
        for i, im in enumerate(images):
            d = im.to(device, non_blocking=True).contiguous()
            rc = _native.lib.datr_normalize_pad_u8_f32(
                d.data_ptr(), int(d.shape[0]), int(d.shape[1]), ctypes.cast(m, ctypes.c_void_p),
                ctypes.cast(s, ctypes.c_void_p), Hp, Wp, int(channels_last),
                batch.data_ptr() + i * slot, mask.data_ptr() + i * Hp * Wp, stream)
            _native.check(rc, "normalize_pad")
            d.record_stream(torch.cuda.current_stream(device))
    padded = any(int(im.shape[0]) != Hp or int(im.shape[1]) != Wp for im in images)
    return NestedTensor(batch, mask, padded)


def collate_fn_da_on_device(batch, device=None, channels_last: bool = True):
    """`collate_fn_da` (/root/reference/util/misc.py:291-300) for items whose images are still uint8
    [H, W, 3] (transforms.da_item): samples = source images followed by target images in one padded
    batch; samples_strong_aug = source images followed by the strongly augmented target images (None
    when the items carry no strong image).  Returns (samples, source_labels, target_labels,
    samples_strong_aug)."""
    source_imgs, source_labels, target_imgs, target_labels, target_imgs_strong_aug = list(zip(*batch))
    samples = collate_uint8_on_device(source_imgs + target_imgs, device=device, channels_last=channels_last)
    samples_strong_aug = None
    if target_imgs_strong_aug[0] is not None:
        samples_strong_aug = collate_uint8_on_device(source_imgs + target_imgs_strong_aug, device=device,
                                                     channels_last=channels_last)
    return samples, source_labels, target_labels, samples_strong_aug


def get_size_with_aspect_ratio(image_size_wh, size, max_size=None):
    """(oh, ow) of the reference's RandomResize (da_transforms.py:88-106): the shorter side becomes
    `size` unless the longer one would exceed `max_size`."""
    w, h = image_size_wh
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def pillow_coeffs(in_size: int, out_size: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the full box
    (src/libImaging/Resample.c): -> (bounds int32 [out, 2] = {first index, count},
    weights int32 [out, ksize], ksize).  Double arithmetic and truncations as in the C source."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


_COEFFS = {}


def _device_coeffs(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _COEFFS:
        if len(_COEFFS) > 512:
            _COEFFS.clear()
        b, k, ks = pillow_coeffs(in_size, out_size)
        _COEFFS[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return _COEFFS[key]


def resize_uint8_on_device(image: torch.Tensor, size_hw, flip: bool = False) -> torch.Tensor:
    """`F.resize(F.hflip(img) if flip else img, size_hw)` of the reference's transforms for a uint8
    [H, W, 3] image: returns the uint8 [oh, ow, 3] device tensor Pillow's bilinear resize produces,
    bit for bit (csrc/resize.hip; the weight tables are cached per (source, target) length)."""
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise ValueError("image must be uint8 [H, W, 3]")
    if not image.is_cuda:
        raise RuntimeError("resize_uint8_on_device: Not implemented on the CPU")
    H, W = int(image.shape[0]), int(image.shape[1])
    oh, ow = int(size_hw[0]), int(size_hw[1])
    src = image.contiguous()
    dst = torch.empty(oh, ow, 3, dtype=torch.uint8, device=image.device)
    tmp = torch.empty(H, max(ow, W), 3, dtype=torch.uint8, device=image.device)
    xb, xk, ksx = _device_coeffs(W, ow, image.device) if ow != W else (None, None, 0)
    yb, yk, ksy = _device_coeffs(H, oh, image.device) if oh != H else (None, None, 0)
    ptr = lambda t: 0 if t is None else t.data_ptr()
    with _native.on_device(image.device):
        rc = _native.lib.datr_resize_bilinear_u8(src.data_ptr(), H, W, int(bool(flip)), ptr(xb), ptr(xk), ksx,
                                                 ptr(yb), ptr(yk), ksy, oh, ow, tmp.data_ptr(), dst.data_ptr(),
                                                 _native.current_stream_ptr(image.device))
    _native.check(rc, "resize_bilinear_u8")
    return dst


def hflip_boxes(boxes_xyxy: torch.Tensor, width: int) -> torch.Tensor:
    """Box update of the reference's hflip (da_transforms.py:74-77)."""
    return boxes_xyxy[:, [2, 1, 0, 3]] * torch.as_tensor([-1, 1, -1, 1], dtype=boxes_xyxy.dtype) \
        + torch.as_tensor([width, 0, width, 0], dtype=boxes_xyxy.dtype)


def resize_boxes(boxes_xyxy: torch.Tensor, old_wh, new_wh) -> torch.Tensor:
    """Box update of the reference's resize (da_transforms.py:122-130)."""
    rw, rh = float(new_wh[0]) / float(old_wh[0]), float(new_wh[1]) / float(old_wh[1])
    return boxes_xyxy * torch.as_tensor([rw, rh, rw, rh], dtype=boxes_xyxy.dtype)
