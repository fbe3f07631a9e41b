"""oracle/msda_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end for oracle/libdatr_oracle.so (the plain-C restatement in
oracle/msda_ref.c) plus a torch `grid_sample` formulation of the same operator.

Who may import this: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg.  datr_amd/ (the product) never imports anything from oracle/.

Reference behaviour (under /root/reference):
  * C functions       -> models/dino/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-403
  * grid_sample form  -> models/dino/ops/functions/ms_deform_attn_func.py:41-61
    (`ms_deform_attn_core_pytorch`, the reference's own "debug/test only" path,
    which is also the comparison target of models/dino/ops/test.py:31-60)
Parity pin: tests/test_oracle_msda.py checks both against tests/golden/msda_*.npz,
which were produced by importing the reference in the build container
(tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdatr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/*.c with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdatr_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        i64 = ctypes.c_int64
        vp = ctypes.c_void_p
        for sfx in ("f32", "f64"):
            f = getattr(_lib, f"datr_oracle_msda_forward_{sfx}")
            f.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, vp]
            f.restype = ctypes.c_int
            b = getattr(_lib, f"datr_oracle_msda_backward_{sfx}")
            b.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, vp, vp, vp]
            b.restype = ctypes.c_int
    return _lib


def _prep(t: torch.Tensor, dtype) -> torch.Tensor:
    return t.detach().to(device="cpu", dtype=dtype).contiguous()


def _dims(value, shapes, loc):
    N, S, M, D = value.shape
    L = shapes.shape[0]
    Lq, P = loc.shape[1], loc.shape[4]
    return N, S, M, D, L, Lq, P


def msda_forward(value, shapes, lsi, loc, attn) -> torch.Tensor:
    """C oracle forward. Tensors may live anywhere; the result is a CPU tensor
    of value's dtype (float32 or float64), shape [N, Lq, M*D]."""
    dt = value.dtype
    assert dt in (torch.float32, torch.float64)
    value, loc, attn = _prep(value, dt), _prep(loc, dt), _prep(attn, dt)
    shapes, lsi = _prep(shapes, torch.int64), _prep(lsi, torch.int64)
    N, S, M, D, L, Lq, P = _dims(value, shapes, loc)
    out = torch.empty(N, Lq, M * D, dtype=dt)
    fn = getattr(lib(), "datr_oracle_msda_forward_" + ("f32" if dt == torch.float32 else "f64"))
    rc = fn(value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(),
            attn.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr())
    assert rc == 0
    return out


def msda_backward(value, shapes, lsi, loc, attn, grad_out):
    """C oracle backward -> (grad_value, grad_loc, grad_attn) CPU tensors."""
    dt = value.dtype
    assert dt in (torch.float32, torch.float64)
    value, loc, attn, grad_out = (_prep(value, dt), _prep(loc, dt), _prep(attn, dt),
                                  _prep(grad_out, dt))
    shapes, lsi = _prep(shapes, torch.int64), _prep(lsi, torch.int64)
    N, S, M, D, L, Lq, P = _dims(value, shapes, loc)
    gv = torch.zeros_like(value)
    gl = torch.empty_like(loc)
    ga = torch.empty_like(attn)
    fn = getattr(lib(), "datr_oracle_msda_backward_" + ("f32" if dt == torch.float32 else "f64"))
    rc = fn(grad_out.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
            loc.data_ptr(), attn.data_ptr(), N, S, M, D, L, Lq, P,
            gv.data_ptr(), gl.data_ptr(), ga.data_ptr())
    assert rc == 0
    return gv, gl, ga


def msda_grid_sample(value, shapes, loc, attn) -> torch.Tensor:
    """torch formulation: per level, bilinear `grid_sample` (zero padding,
    align_corners=False) of the head-major value map at 2*loc-1, weighted by the
    attention weights and summed over (level, point).  Differentiable, so
    autograd through it is the backward oracle for arbitrary dtypes."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    hw = [(int(h), int(w)) for h, w in shapes.tolist()]
    per_level = value.split([h * w for h, w in hw], dim=1)
    grid = 2.0 * loc - 1.0
    sampled = []
    for lvl, (h, w) in enumerate(hw):
        # [N, h*w, M, D] -> [N*M, D, h, w]
        fmap = per_level[lvl].permute(0, 2, 3, 1).reshape(N * M, D, h, w)
        # [N, Lq, M, P, 2] -> [N*M, Lq, P, 2]
        g = grid[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        sampled.append(F.grid_sample(fmap, g, mode="bilinear", padding_mode="zeros",
                                     align_corners=False))           # [N*M, D, Lq, P]
    sampled = torch.stack(sampled, dim=3)                              # [N*M, D, Lq, L, P]
    w_ = attn.permute(0, 2, 1, 3, 4).reshape(N * M, 1, Lq, L, P)
    out = (sampled * w_).sum(dim=(3, 4))                               # [N*M, D, Lq]
    return out.reshape(N, M * D, Lq).transpose(1, 2).contiguous()


def level_start_index(shapes: torch.Tensor) -> torch.Tensor:
    areas = shapes[:, 0] * shapes[:, 1]
    return torch.cat([areas.new_zeros(1), areas.cumsum(0)[:-1]])


def random_inputs(N, Lq, M, D, shapes, P, seed=3, dtype=torch.float32, loc_range=(0.0, 1.0)):
    """Seeded inputs in the style of the reference's op test
    (models/dino/ops/test.py:31-37): value = U[0,1)*0.01, loc = U[lo,hi),
    attention weights normalised over (L, P)."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shapes, dtype=torch.int64)
    L = shapes.shape[0]
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = torch.rand(N, S, M, D, generator=g, dtype=torch.float64) * 0.01
    lo, hi = loc_range
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * (hi - lo) + lo
    attn = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64) + 1e-5
    attn = attn / attn.sum(dim=(-1, -2), keepdim=True)
    return (value.to(dtype), shapes, level_start_index(shapes), loc.to(dtype), attn.to(dtype))
