"""Device-side tail of the input pipeline (first piece).

The reference's data loader produces, per image, a normalised float CHW tensor on the host
(`T.ToTensor` + `T.Normalize`, /root/reference/datasets/da_transforms.py:250-276) and
`collate_fn_da` -> `nested_tensor_from_tensor_list` (/root/reference/util/misc.py:291-300,
:387-409) pads them into one batch with a bool mask.  `collate_uint8_on_device` takes the
uint8 HWC images instead (what the decoder / resize / flip stages hand over), moves a quarter
of the bytes across PCIe, and builds the padded batch -- directly in the backbone's NHWC layout
if asked -- and the mask in one kernel per image (csrc/preprocess.hip).  Same values bit for bit.
Resize, flip and the strong-augmentation ops stay on the host this round.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

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
    with torch.cuda.device(device):
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
