"""Multi-scale deformable attention: Python mirror of the reference's op package.

Mirrors, name for name and argument for argument:
  * the pybind11 module `MultiScaleDeformableAttention`
    (`ms_deform_attn_forward`, `ms_deform_attn_backward`;
    /root/reference/models/dino/ops/src/vision.cpp:13-16, src/ms_deform_attn.h:21-60,
    src/cuda/ms_deform_attn_cuda.cu:20-153),
  * `MSDeformAttnFunction` (/root/reference/models/dino/ops/functions/ms_deform_attn_func.py:21-38),
  * the `MSDeformAttn` module (/root/reference/models/dino/ops/modules/ms_deform_attn.py:31-126),
with the compute done by libdatr_hip.so through the C ABI in include/datr_hip.h.  Error
behaviour follows the reference: non-contiguous tensors raise RuntimeError, CPU tensors raise
"Not implemented on the CPU" (ms_deform_attn.h:38,60), a batch that is not a multiple of
min(batch, im2col_step) raises (ms_deform_attn_cuda.cu:50-52).
"""
from __future__ import annotations

import math
import warnings
import weakref

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native
from .fused import FastLinear, _amp_bwd, _amp_fwd, linear as fast_linear

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward", "MSDeformAttnFunction",
           "MSDeformAttn"]


def _require(t: torch.Tensor, name: str) -> None:
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
    if not t.is_cuda:
        raise RuntimeError("Not implemented on the CPU")


def _dims(value, spatial_shapes, sampling_loc, im2col_step):
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    step = min(batch, im2col_step)
    if step <= 0 or batch % step != 0:
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    return batch, spatial_size, num_heads, channels, num_levels, num_query, num_point


def _suffix(value: torch.Tensor) -> str:
    if value.dtype == torch.float32:
        return "f32"
    if value.dtype == torch.float64:
        return "f64"
    raise RuntimeError(f"ms_deform_attn: unsupported dtype {value.dtype} (float32/float64 only)")


def _as_int64(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int64 else t.to(torch.int64)


_HOST_META = {}
# Encoder calls (Lq == S, D == 32, L == P == 4) take the pyramid-region forward of
# csrc/msda_fwd_pyr.hip (levels 1..3 gathered out of LDS, level 0 through the vector-memory
# path); everything else the row kernel.  DATR_MSDA_PYR_FWD=0 forces the row kernel (A/B runs).
TILED_BACKWARD_MIN_LQ = int(__import__("os").environ.get("DATR_MSDA_TILED_BWD_MIN_LQ", "64"))
MERGE_QUERY_PROJECTIONS = __import__("os").environ.get("DATR_MERGE_QPROJ", "1") != "0"   # A/B switch
VALUE_PROJ_BATCH = __import__("os").environ.get("DATR_VALUE_PROJ_BATCH", "1") != "0"     # A/B switch
_EUNSUPPORTED = -2                              # DATR_EUNSUPPORTED of include/datr_hip.h
PYR_FORWARD = __import__("os").environ.get("DATR_MSDA_PYR_FWD", "1") != "0"


def _host_meta(shapes: torch.Tensor, lsi: torch.Tensor):
    """Host copies (numpy int64) of the two small geometry tensors, cached so that a training
    loop pays the device->host copy once per geometry, not per call.  The key is the storage
    address + version of both tensors; the entry HOLDS the two tensors, so their storage cannot
    be freed and handed to a different geometry while the key is in the cache (an address is an
    identity only for as long as its owner lives -- whoever else caches the device tensors,
    e.g. datr_amd.transformer._level_meta, may drop them at any time)."""
    key = (shapes.data_ptr(), shapes._version, lsi.data_ptr(), lsi._version, shapes.shape[0])
    hit = _HOST_META.get(key)
    if hit is None:
        if len(_HOST_META) > 256:
            _HOST_META.clear()
        hit = _HOST_META[key] = (shapes.cpu().numpy().copy(), lsi.cpu().numpy().copy(),
                                 (shapes, lsi))
    return hit[0], hit[1]


def pyramid_plan(spatial_shapes, level_start_index, N, M, D, P, envelope=None):
    """What the pyramid-region kernels would do for an encoder call (Lq == S) of this geometry,
    without launching anything: dict(forward=bool, grid=(nRy, nRx), phases, tasks_per_wave,
    workgroups_per_image, fill_kib, backward=bool, backward_grid).  `datr_msda_pyramid_plan`."""
    import numpy as np
    shapes, lsi = _as_int64(spatial_shapes), _as_int64(level_start_index)
    sh_host, ls_host = (shapes.cpu().numpy().copy(), lsi.cpu().numpy().copy()) if not shapes.is_cuda \
        else _host_meta(shapes, lsi)
    S = int((sh_host[:, 0] * sh_host[:, 1]).sum())
    info = np.zeros(16, dtype=np.int32)
    env = None if envelope is None else np.ascontiguousarray(envelope, dtype=np.float32)
    assert env is None or env.shape == (8, 4, 4)
    rc = _native.lib.datr_msda_pyramid_plan(sh_host.ctypes.data, ls_host.ctypes.data, N, S, M, D,
                                            sh_host.shape[0], S, P, None if env is None else env.ctypes.data,
                                            info.ctypes.data)
    _native.check(rc, "pyramid_plan")
    return {"forward": bool(info[0]), "grid": (int(info[1]), int(info[2])), "phases": int(info[3]),
            "tasks_per_wave": int(info[4]), "workgroups_per_image": int(info[5]), "fill_kib": int(info[6]),
            "largest_phase_rows": int(info[7]), "phased": bool(info[11]), "config": int(info[12]), "backward": bool(info[8]),
            "backward_grid": (int(info[9]), int(info[10]))}


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step: int, route: int = 0, envelope=None):
    """-> Tensor [N, Lq, M*D]  (same contract as MSDA.ms_deform_attn_forward).
    route (see OffsetMonitor): 0 = kernel by geometry; > 0 = no pyramid-region kernel.
    envelope: optional numpy float32 [8, 4, 4] = per (head, level) {oy_lo, oy_hi, ox_lo, ox_hi}
    in pixels, how far the samples lie from their reference points (a performance hint for the
    phased pyramid kernel's window sizes; None = symmetric 4.5 px)."""
    for t, nme in ((value, "value"), (spatial_shapes, "spatial_shapes"),
                   (level_start_index, "level_start_index"), (sampling_loc, "sampling_loc"),
                   (attn_weight, "attn_weight")):
        _require(t, nme)
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    sfx = _suffix(value)
    if sampling_loc.dtype != value.dtype:
        sampling_loc = sampling_loc.to(value.dtype)
    if attn_weight.dtype != value.dtype:
        attn_weight = attn_weight.to(value.dtype)
    shapes, lsi = _as_int64(spatial_shapes), _as_int64(level_start_index)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    if envelope is not None:                 # read by the planner through a raw pointer
        import numpy as np
        envelope = np.ascontiguousarray(envelope, dtype=np.float32)
        if envelope.shape != (8, 4, 4):
            raise ValueError(f"offset envelope must be [8, 4, 4], got {envelope.shape}")
    with _native.on_device(value.device):
        stream = _native.current_stream_ptr(value.device)
        if sfx == "f32" and D == 32 and Lq == S and L == 4 and P == 4 and PYR_FORWARD and route == 0:
            # encoder self-attention: pyramid-region forward, coarse-level windows staged in LDS
            sh_host, ls_host = _host_meta(shapes, lsi)
            rc = _native.lib.datr_msda_forward_pyramid_f32(
                value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), sh_host.ctypes.data,
                ls_host.ctypes.data, None if envelope is None else envelope.ctypes.data,
                sampling_loc.data_ptr(), attn_weight.data_ptr(),
                N, S, M, D, L, Lq, P, out.data_ptr(), stream)
        else:
            fn = getattr(_native.lib, f"datr_msda_forward_{sfx}")
            rc = fn(value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), sampling_loc.data_ptr(),
                    attn_weight.data_ptr(), N, S, M, D, L, Lq, P, out.data_ptr(), stream)
    _native.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step: int, route: int = 0, envelope=None, grad_value_out=None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight].
    route: 0 = kernel by geometry; 1 = no pyramid-region kernel; 2 = the row kernel.
    envelope: the forward's measured offset envelope (numpy float32 [8, 4, 4]) or None; sizes the windows of
    the encoder calls' pyramid-region kernel, never changes a result.
    grad_value_out: optional [N, S, M, D] float32 view whose pixel rows may sit further apart than M * D
    floats (a column slice of a buffer several calls share, see value_projections): grad_value is written
    there -- directly by the decoder calls' kernel, through a copy otherwise -- and returned."""
    for t, nme in ((value, "value"), (spatial_shapes, "spatial_shapes"),
                   (level_start_index, "level_start_index"), (sampling_loc, "sampling_loc"),
                   (attn_weight, "attn_weight"), (grad_output, "grad_output")):
        _require(t, nme)
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    sfx = _suffix(value)
    shapes, lsi = _as_int64(spatial_shapes), _as_int64(level_start_index)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    if grad_value_out is not None and sfx == "f32" and D == 32 and route == 0 and Lq != S \
            and tuple(grad_value_out.shape) == (N, S, M, D) and grad_value_out.stride(3) == 1 \
            and grad_value_out.stride(2) == D and grad_value_out.stride(0) == S * grad_value_out.stride(1):
        sh_host, ls_host = _host_meta(shapes, lsi)
        with _native.on_device(value.device):
            rc = _native.lib.datr_msda_backward_strided_f32(
                grad_output.data_ptr(), value.data_ptr(), sh_host.ctypes.data, ls_host.ctypes.data,
                sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, grad_value_out.data_ptr(),
                grad_value_out.stride(1), grad_loc.data_ptr(), grad_attn.data_ptr(),
                _native.current_stream_ptr(value.device))
        if rc != _EUNSUPPORTED:
            _native.check(rc, "ms_deform_attn_backward (strided)")
            return [grad_value_out, grad_loc, grad_attn]
    grad_value = torch.empty_like(value)            # zero-filled by the library on the stream
    with _native.on_device(value.device):
        stream = _native.current_stream_ptr(value.device)
        if sfx == "f32" and D == 32 and Lq >= TILED_BACKWARD_MIN_LQ and route < 2:
            # geometry-dispatched backward (pyramid regions / owner-computes / query tiles): needs
            # the geometry on the host
            sh_host, ls_host = _host_meta(shapes, lsi)
            if route == 0:
                env = None
                if envelope is not None:
                    import numpy as np
                    env = np.ascontiguousarray(envelope, dtype=np.float32)
                    assert env.shape == (8, 4, 4)
                rc = _native.lib.datr_msda_backward_pyramid_f32(
                    grad_output.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
                    sh_host.ctypes.data, ls_host.ctypes.data, 0 if env is None else env.ctypes.data,
                    sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                    grad_loc.data_ptr(), grad_attn.data_ptr(), stream)
            else:
                rc = _native.lib.datr_msda_backward_query_tiled_f32(
                    grad_output.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
                    sh_host.ctypes.data, ls_host.ctypes.data, sampling_loc.data_ptr(),
                    attn_weight.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                    grad_loc.data_ptr(), grad_attn.data_ptr(), stream)
        else:
            fn = getattr(_native.lib, f"datr_msda_backward_{sfx}")
            rc = fn(grad_output.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(),
                    sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P,
                    grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), stream)
    _native.check(rc, "ms_deform_attn_backward")
    if grad_value_out is not None:
        grad_value_out.copy_(grad_value)
        grad_value = grad_value_out
    return [grad_value, grad_loc, grad_attn]


def ms_deform_attn_backward_query_grad(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                       grad_output, im2col_step: int, envelope=None):
    """The encoder calls' backward for the MODULE: -> (grad_value, grad_query) with grad_query [N, Lq, M * 48] the
    gradient of the merged (offsets | logits) query projection -- grad_sampling_loc and the softmax backward of
    grad_attn_weight already in that projection's column layout (include/datr_hip.h,
    datr_msda_backward_pyramid_query_f32) --, or None when the library does not cover the shape (the caller then runs
    ms_deform_attn_backward and the prologue's own backward)."""
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    if not (_suffix(value) == "f32" and D == 32 and Lq == S and M == 8 and L == 4 and P == 4
            and sampling_loc.dtype == value.dtype and attn_weight.dtype == value.dtype):
        return None
    for t, nme in ((value, "value"), (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight"),
                   (grad_output, "grad_output")):
        _require(t, nme)
    shapes, lsi = _as_int64(spatial_shapes), _as_int64(level_start_index)
    sh_host, ls_host = _host_meta(shapes, lsi)
    env = None
    if envelope is not None:
        import numpy as np
        env = np.ascontiguousarray(envelope, dtype=np.float32)
        assert env.shape == (8, 4, 4)
    grad_value = torch.empty_like(value)            # zero-filled by the library on the stream
    grad_query = torch.empty((N, Lq, M * 48), dtype=value.dtype, device=value.device)
    with _native.on_device(value.device):
        rc = _native.lib.datr_msda_backward_pyramid_query_f32(
            grad_output.data_ptr(), value.data_ptr(), sh_host.ctypes.data, ls_host.ctypes.data,
            0 if env is None else env.ctypes.data, sampling_loc.data_ptr(), attn_weight.data_ptr(),
            N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_query.data_ptr(),
            _native.current_stream_ptr(value.device))
    if rc == _EUNSUPPORTED:
        return None
    _native.check(rc, "ms_deform_attn_backward_query_grad")
    return grad_value, grad_query


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step, route=0, envelope=None, grad_slot=None):
        ctx.im2col_step = im2col_step
        ctx.grad_slot = grad_slot                    # (GradSlab, index): where grad_value goes (value_projections)
        ctx.route = int(route)
        kw = {"route": ctx.route} if ctx.route else {}
        if envelope is not None:
            # the planner reads 8 x 4 x 4 float32 values through a raw pointer: normalise once, for both directions
            import numpy as np
            envelope = np.ascontiguousarray(envelope, dtype=np.float32)
            if envelope.shape != (8, 4, 4):
                raise ValueError(f"offset envelope must be [8, 4, 4], got {envelope.shape}")
            kw["envelope"] = envelope
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                        sampling_locations, attention_weights, ctx.im2col_step, **kw)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        ctx.envelope = envelope
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        kw = {"route": ctx.route} if ctx.route else {}
        if ctx.envelope is not None:
            kw["envelope"] = ctx.envelope
        if ctx.grad_slot is not None and value.dim() == 4:
            slab, i = ctx.grad_slot
            kw["grad_value_out"] = slab.slot(i, value)
        grad_value, grad_loc, grad_attn = ms_deform_attn_backward(
            value, shapes, lsi, loc, attn, grad_output.contiguous(), ctx.im2col_step, **kw)
        return grad_value, None, None, grad_loc, grad_attn, None, None, None, None


class GradSlab:
    """One [N, S, n * C] buffer for the value gradients of n attention calls on the same memory: call i
    writes columns i C .. (i + 1) C (its kernel takes the row stride), so that the n value projections'
    data gradient, weight gradients and bias gradients are ONE GEMM / GEMM / column sum over the buffer
    instead of n of each plus n - 1 adds over the token tensor."""

    def __init__(self, n: int):
        self.n, self.buf = n, None

    def slot(self, i: int, value: torch.Tensor) -> torch.Tensor:
        N, S, M, D = value.shape
        C = M * D
        if self.buf is None or tuple(self.buf.shape) != (N, S, self.n * C) or self.buf.device != value.device:
            self.buf = torch.empty(N, S, self.n * C, device=value.device, dtype=torch.float32)
        return self.buf[:, :, i * C:(i + 1) * C].view(N, S, M, D)

    def holds(self, grads, C: int) -> bool:
        b = self.buf
        if b is None or len(grads) != self.n:
            return False
        N, S, _ = b.shape
        want = (S * self.n * C, self.n * C, 1)
        return all(g is not None and tuple(g.shape) == (N, S, C) and g.stride() == want
                   and g.data_ptr() == b.data_ptr() + 4 * i * C for i, g in enumerate(grads))


class _ValueProjN(Function):
    """[memory @ W_i^T + b_i for i in range(n)]: the value projections of the n decoder layers
    (/root/reference/models/dino/ops/modules/ms_deform_attn.py:96-100 inside
    deformable_transformer.py:880-900, all on the encoder's memory).  Forward: n GEMMs.  Backward: when the
    attention calls left their value gradients in the shared GradSlab, ONE data-gradient GEMM (K = n C; its
    reduction replaces n - 1 adds over the 91 MB token tensor), ONE weight-gradient GEMM and one column sum."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, memory, slab, *wb):
        n = len(wb) // 2
        ws, bs = wb[:n], wb[n:]
        C = memory.shape[-1]
        m2 = memory.reshape(-1, C)
        if m2.shape[0] >= 16384 and m2.is_contiguous() and C % 32 == 0 and all(w.shape[0] % 4 == 0 and w.is_contiguous() for w in ws):
            # own MFMA GEMM with the bias in the epilogue (csrc/gemm_f32.hip): 111 us per projection at the
            # step's 88 892 rows against 147 us for the library's pick for this call
            from . import gemm
            outs = tuple(gemm.gemm_nt(m2, w, shift=b.contiguous()).view(*memory.shape[:-1], w.shape[0]) for w, b in zip(ws, bs))
        else:
            outs = tuple(torch.addmm(b, m2, w.t()).view(*memory.shape[:-1], w.shape[0]) for w, b in zip(ws, bs))
        ctx.save_for_backward(memory, *ws)
        ctx.slab, ctx.n = slab, n
        return outs

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, *gs):
        from .fused import _dgrad, _wgrad_mm, column_sums
        memory, *ws = ctx.saved_tensors
        n, slab = ctx.n, ctx.slab
        C = memory.shape[-1]
        m2 = memory.reshape(-1, C)
        need = ctx.needs_input_grad
        if all(w.shape[0] == C for w in ws) and slab.holds(gs, C):
            buf, slab.buf = slab.buf, None                # the next step gets a fresh buffer
            g2 = buf.view(-1, n * C)
            dmem = _dgrad(g2, torch.cat(ws, 0)).view_as(memory) if need[0] else None
            dws = _wgrad_mm(g2, m2).split(C, 0)
            dbs = column_sums(g2).split(C, 0)
        else:
            dmem, dws, dbs = None, [], []
            for g, w in zip(gs, ws):
                if g is None:
                    dws.append(None); dbs.append(None)
                    continue
                g2 = g.reshape(-1, w.shape[0])
                g2 = g2 if g2.is_contiguous() else g2.contiguous()
                if need[0]:
                    dmem = g2.mm(w) if dmem is None else dmem.addmm_(g2, w)
                dws.append(g2.t().mm(m2))
                dbs.append(column_sums(g2))
            dmem = None if dmem is None else dmem.view_as(memory)
        return (dmem, None, *dws, *dbs)


def value_projections(memory: torch.Tensor, attns):
    """(values, slab) for attention modules that all read `memory` ([N, S, C]): values[i] =
    attns[i].value_proj(memory), computed by one autograd node whose backward is batched over the modules
    (_ValueProjN); hand values[i] and (slab, i) to attns[i].forward.  None when that does not apply."""
    if not (VALUE_PROJ_BATCH and 2 <= len(attns) <= 8 and memory.is_cuda and memory.dtype == torch.float32
            and memory.dim() == 3 and memory.is_contiguous() and torch.is_grad_enabled()
            and all(isinstance(a, MSDeformAttn) and a.value_proj.bias is not None
                    and a.value_proj.weight.shape == (memory.shape[-1], memory.shape[-1])
                    and a.d_model // a.n_heads == 32 for a in attns)):
        return None
    slab = GradSlab(len(attns))
    values = _ValueProjN.apply(memory, slab, *[a.value_proj.weight for a in attns],
                               *[a.value_proj.bias for a in attns])
    return values, slab


def _is_power_of_2(n: int) -> bool:
    if not isinstance(n, int) or n < 0:
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return n != 0 and (n & (n - 1)) == 0


class _SplitLast(Function):
    """x[..., :n], x[..., n:] as views; backward is ONE concatenation of the two gradients
    (autograd's slice backward would zero-fill the full tensor twice and add)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = n, x.shape
        return x[..., :n], x[..., n:]

    @staticmethod
    def backward(ctx, ga, gb):
        ref = ga if ga is not None else gb
        if ga is None:
            ga = ref.new_zeros(*ctx.shape[:-1], ctx.n)
        if gb is None:
            gb = ref.new_zeros(*ctx.shape[:-1], ctx.shape[-1] - ctx.n)
        return torch.cat([ga, gb], -1), None


def _prologue_forward(both, ref):
    rows = both.numel() // 384
    loc = torch.empty(*both.shape[:-1], 8, 4, 4, 2, device=both.device, dtype=torch.float32)
    attn = torch.empty(*both.shape[:-1], 8, 4, 4, device=both.device, dtype=torch.float32)
    with _native.on_device(both.device):
        rc = _native.lib.datr_msda_prologue_forward_f32(
            both.data_ptr(), ref.data_ptr(), rows, ref.shape[-1], loc.data_ptr(), attn.data_ptr(),
            _native.current_stream_ptr(both.device))
    _native.check(rc, "msda_prologue_forward")
    return loc, attn


def _prologue_backward(d_loc, d_attn, attn, ref, shape):
    d_loc = torch.zeros_like(attn).unsqueeze(-1).expand(*attn.shape, 2).contiguous() \
        if d_loc is None else d_loc.contiguous()
    d_attn = torch.zeros_like(attn) if d_attn is None else d_attn.contiguous()
    d_both = torch.empty(shape, device=attn.device, dtype=torch.float32)
    with _native.on_device(attn.device):
        rc = _native.lib.datr_msda_prologue_backward_f32(
            d_loc.data_ptr(), d_attn.data_ptr(), attn.data_ptr(), ref.data_ptr(),
            d_both.numel() // 384, ref.shape[-1], d_both.data_ptr(),
            _native.current_stream_ptr(attn.device))
    _native.check(rc, "msda_prologue_backward")
    return d_both


class _Prologue(Function):
    """(sampling locations, attention weights) from the merged query projection in one launch
    each way (csrc/msda_prologue.hip); 8 heads x 4 levels x 4 points, reference points without
    gradient."""

    @staticmethod
    def forward(ctx, both, ref):
        both, ref = both.contiguous(), ref.contiguous()
        loc, attn = _prologue_forward(both, ref)
        ctx.save_for_backward(attn, ref)
        ctx.shape = both.shape
        return loc, attn

    @staticmethod
    @once_differentiable
    def backward(ctx, d_loc, d_attn):
        attn, ref = ctx.saved_tensors
        return _prologue_backward(d_loc, d_attn, attn, ref, ctx.shape), None


class _PrologueMSDA(Function):
    """_Prologue and MSDeformAttnFunction of an encoder self-attention call (2-d reference points, the division by
    (W_l, H_l) folded into the projection: locations = reference + offsets) as ONE node, for its backward: the
    LDS-window kernel that computes grad_sampling_loc / grad_attn_weight writes the query projection's gradient rows
    itself (softmax backward included), where the two nodes wrote 136 MB per call for a separate pass to re-arrange
    (ms_deform_attn.py:94-113).  Returns (output, sampling locations); the locations carry no gradient (the offset
    monitor reads them)."""

    @staticmethod
    def forward(ctx, both, ref, value, shapes, lsi, im2col_step, route, envelope):
        both, ref = both.contiguous(), ref.contiguous()
        loc, attn = _prologue_forward(both, ref)
        ctx.im2col_step, ctx.route, ctx.shape = im2col_step, int(route), both.shape
        kw = {"route": ctx.route} if ctx.route else {}
        if envelope is not None:
            import numpy as np
            envelope = np.ascontiguousarray(envelope, dtype=np.float32)
            if envelope.shape != (8, 4, 4):
                raise ValueError(f"offset envelope must be [8, 4, 4], got {envelope.shape}")
            kw["envelope"] = envelope
        ctx.envelope = envelope
        out = ms_deform_attn_forward(value, shapes, lsi, loc, attn, im2col_step, **kw)
        ctx.save_for_backward(value, shapes, lsi, loc, attn, ref)
        ctx.mark_non_differentiable(loc)
        ctx.set_materialize_grads(False)        # no 91-MB zero gradient for the locations
        return out, loc

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, _unused):
        value, shapes, lsi, loc, attn, ref = ctx.saved_tensors
        if grad_output is None:
            return (None,) * 8
        grad_output = grad_output.contiguous()
        if ctx.route == 0 and QUERY_GRAD_BACKWARD:
            kw = {} if ctx.envelope is None else {"envelope": ctx.envelope}
            done = ms_deform_attn_backward_query_grad(value, shapes, lsi, loc, attn, grad_output, ctx.im2col_step, **kw)
            if done is not None:
                return done[1].view(ctx.shape), None, done[0], None, None, None, None, None
        kw = {"route": ctx.route} if ctx.route else {}
        if ctx.envelope is not None:
            kw["envelope"] = ctx.envelope
        grad_value, grad_loc, grad_attn = ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output,
                                                                  ctx.im2col_step, **kw)
        return _prologue_backward(grad_loc, grad_attn, attn, ref, ctx.shape), None, grad_value, None, None, None, None, None


QUERY_GRAD_BACKWARD = __import__("os").environ.get("DATR_MSDA_QUERY_GRAD", "1") != "0"   # A/B switch
FUSED_PROLOGUE = __import__("os").environ.get("DATR_FUSED_PROLOGUE", "1") != "0"      # A/B switch
_INV_WH = {}


def _inverse_wh(spatial_shapes: torch.Tensor, n_heads: int, n_points: int) -> torch.Tensor:
    """[n_heads * L * n_points * 2] vector of 1/W_l, 1/H_l in the layout of the sampling_offsets
    output (head, level, point, xy); cached per geometry tensor.  As in `_host_meta`, the entry
    keeps the geometry tensor alive, so its address cannot be recycled for another pyramid while
    it serves as the key."""
    key = (spatial_shapes.data_ptr(), spatial_shapes._version, str(spatial_shapes.device), n_heads,
           n_points, tuple(spatial_shapes.shape))
    hit = _INV_WH.get(key)
    if hit is None:
        if len(_INV_WH) > 64:
            _INV_WH.clear()
        wh = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1).to(torch.float32)
        inv = (1.0 / wh)[None, :, None, :].expand(n_heads, -1, n_points, -1).reshape(-1).contiguous()
        hit = _INV_WH[key] = (inv, spatial_shapes)
    return hit[0]


class _StackLinear(torch.autograd.Function):
    """(w, b) = ([diag(scale) wa ; wb], [scale * ba ; bb]) in one launch (csrc/stack_linear.hip); backward: the
    gradients of wb / bb (and of wa / ba without a scale) are row slices of the incoming ones, the scaled block
    takes one launch."""

    @staticmethod
    def forward(ctx, wa, ba, wb, bb, scale):
        Ra, C = wa.shape
        Rb = wb.shape[0]
        w = torch.empty(Ra + Rb, C, device=wa.device, dtype=torch.float32)
        b = torch.empty(Ra + Rb, device=wa.device, dtype=torch.float32)
        with _native.on_device(wa.device):
            rc = _native.lib.datr_stack_linear_forward_f32(
                wa.data_ptr(), ba.data_ptr(), wb.data_ptr(), bb.data_ptr(), 0 if scale is None else scale.data_ptr(),
                Ra, Rb, C, w.data_ptr(), b.data_ptr(), _native.current_stream_ptr(wa.device))
        _native.check(rc, "stack_linear_forward")
        ctx.scale = scale
        ctx.Ra = Ra
        return w, b

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dw, db):
        Ra, scale = ctx.Ra, ctx.scale
        dw = dw if dw.is_contiguous() else dw.contiguous()
        db = db if db.is_contiguous() else db.contiguous()
        if scale is None:
            return dw[:Ra], db[:Ra], dw[Ra:], db[Ra:], None
        dwa = torch.empty(Ra, dw.shape[1], device=dw.device, dtype=torch.float32)
        dba = torch.empty(Ra, device=dw.device, dtype=torch.float32)
        with _native.on_device(dw.device):
            rc = _native.lib.datr_stack_linear_backward_f32(dw.data_ptr(), db.data_ptr(), scale.data_ptr(), Ra,
                                                            dw.shape[1], dwa.data_ptr(), dba.data_ptr(),
                                                            _native.current_stream_ptr(dw.device))
        _native.check(rc, "stack_linear_backward")
        return dwa, dba, dw[Ra:], db[Ra:], None


def stack_linear(wa, ba, wb, bb, scale=None):
    """Weights and biases of two linear layers on the same input, stacked for one GEMM; rows of the first layer
    optionally scaled (see _StackLinear).  Device float32 contiguous parameters take the one-launch kernel."""
    if wa.is_cuda and all(t.dtype == torch.float32 and t.is_contiguous() for t in (wa, ba, wb, bb)) \
            and wa.shape[1] % 4 == 0 and (scale is None or (scale.dtype == torch.float32 and scale.is_contiguous())):
        return _StackLinear.apply(wa, ba, wb, bb, scale)
    if scale is not None:
        wa, ba = wa * scale[:, None], ba * scale
    return torch.cat([wa, wb], 0), torch.cat([ba, bb], 0)


class OffsetMonitor:
    """Watches where an encoder layer's samples fall relative to their queries and steers its MSDA
    calls: the ROUTE (which kernel family) and the ENVELOPE (window sizes of the phased pyramid
    forward, csrc/msda_fwd_pyr2.hip).

    Every `every`-th call, on a sub-sample of the queries, the displacement d of every sample from
    its query's own pixel centre (mapped into the sampled level) is measured in pixels of that level:
      * envelope[m, l] = the 0.5 % .. 99.5 % range of d_y and d_x over the samples of head m in
        level l, widened by 0.25 px and clipped to +-8 px.  A DINO encoder's heads look in one
        direction each (ring initialisation, ops/modules/ms_deform_attn.py:59-68), so these boxes
        are a fraction of the symmetric halo; a padded batch's valid-ratio shift of the reference
        points (deformable_transformer.py:524-533) is part of d and widens them as needed.
      * fraction = share of samples with |d| > 4.5 px in x or y; route: f < 0.25 -> 0 (pyramid
        kernels), f < 0.62 -> 1 (row forward, query-tiled backward), else 2 (row kernels) --
        N = 4 encoder call at 1333x800, forward / backward in us for offsets ~ N(0, s px)
        (round 2 kernels, tools/bench_msda.py):
                                 s = 1.5   pyramid 180 /  994    rows+tiled 239 / 1160
                                 s = 4             327 / 2788               249 / 2462
                                 uniform           530 / 51238              272 / 11461   rows 272 / 4668
    The numbers travel to pinned memory without a host synchronisation and are applied by the call
    `LAG` calls later -- a fixed count, so the step at which a route or window plan changes does not
    depend on host / GPU timing and runs are reproducible (the event wait then returns at once).
    Results never depend on either: out-of-window samples take the kernels' slow paths."""

    HALO_PX = 4.5
    LAG = 2                 # calls between a measurement and its use
    MARGIN_PX = 0.25
    CLIP_PX = 8.0
    GRID_PX = float(__import__("os").environ.get("DATR_MSDA_ENVELOPE_GRID", "0.0625"))   # envelopes are rounded outward to this grid (0.25 px cost a window pixel: +0.28 ms per step) ...
    SHRINK_PX = 0.75        # ... and only replaced when they grow, or shrink by at least this much somewhere

    def __init__(self, every: int = 50):
        self.every, self.calls, self.route, self.fraction = every, 0, 0, 0.0
        self.envelope = None                     # numpy float32 [8, 4, 4] once measured
        self._pending = None
        self._host = None
        self._centres = {}

    @staticmethod
    def route_for(fraction: float) -> int:
        return 0 if fraction < 0.25 else (1 if fraction < 0.62 else 2)

    def poll(self):
        if self._pending is not None and self.calls + 1 >= self._pending[2] + self.LAG:   # this is call no. calls + 1
            host, ev, _ = self._pending
            ev.synchronize()
            self.fraction = float(host[0])
            self.route = self.route_for(self.fraction)
            if host.numel() == 1 + 128:
                import numpy as np
                env = host[1:].numpy().reshape(8, 4, 4).copy()
                # Outward to the GRID_PX grid, and KEEP the current envelope while the new one fits inside it
                # and is not much tighter: the measurement is a quantile of a random sub-sample, and an
                # envelope that differs in the last digit is a new plan for the library (a grid search on the
                # host, possibly another kernel variant's first launch: measured 130 ms for the step that
                # adopted a re-measured, practically identical envelope).
                q = self.GRID_PX
                env[..., 0::2] = np.floor(env[..., 0::2] / q) * q
                env[..., 1::2] = np.ceil(env[..., 1::2] / q) * q
                cur = self.envelope
                if cur is None or not (np.all(env[..., 0::2] >= cur[..., 0::2]) and np.all(env[..., 1::2] <= cur[..., 1::2])
                                       and np.all(env[..., 0::2] - cur[..., 0::2] < self.SHRINK_PX)
                                       and np.all(cur[..., 1::2] - env[..., 1::2] < self.SHRINK_PX)):
                    self.envelope = np.ascontiguousarray(env.astype(np.float32))
            self._pending = None
        return self.route

    def _query_centres(self, shapes_host, stride, device):
        """[S', 2] (x, y) pixel centres of every `stride`-th pyramid pixel, normalised to [0, 1]."""
        key = (tuple(map(tuple, shapes_host.tolist())), stride, str(device))
        hit = self._centres.get(key)
        if hit is None:
            refs = []
            for h, w in shapes_host.tolist():
                ys, xs = torch.meshgrid((torch.arange(h, dtype=torch.float32) + 0.5) / h,
                                        (torch.arange(w, dtype=torch.float32) + 0.5) / w, indexing="ij")
                refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
            hit = self._centres[key] = torch.cat(refs, 0)[::stride].contiguous().to(device)
            if len(self._centres) > 8:
                self._centres = {key: hit}
        return hit

    def measure(self, locations: torch.Tensor, shapes_host) -> torch.Tensor:
        """Device tensor [1 + 128] (or [1]): fraction beyond the halo, then the envelope."""
        N, S, M, L, P, _ = locations.shape
        stride = max(1, S // 512)
        ref = self._query_centres(shapes_host, stride, locations.device)
        # device constants are built once per geometry: torch.tensor(list, device=...) is a pageable
        # host-to-device copy, i.e. the host waits for the stream -- six of them (one per encoder layer)
        # cost the measuring step 12 ms of device idle time
        ckey = ("wh", tuple(map(tuple, shapes_host.tolist())), str(locations.device))
        consts = self._centres.get(ckey)
        if consts is None:
            consts = self._centres[ckey] = (
                torch.tensor([[w, h] for h, w in shapes_host.tolist()], dtype=torch.float32,
                             device=locations.device).view(1, 1, 1, L, 1, 2),
                torch.tensor([0.005, 0.995], dtype=torch.float32, device=locations.device))
        wh, quantiles = consts
        d = (locations.detach()[:, ::stride] - ref.view(1, -1, 1, 1, 1, 2)) * wh    # px, (x, y)
        far = (d.abs() > self.HALO_PX).any(-1).float().mean().view(1)
        if (M, L) != (8, 4):
            return far
        v = d.permute(2, 3, 5, 0, 1, 4).reshape(M, L, 2, -1)                    # [m, l, (x, y), samples]
        q = torch.quantile(v, quantiles, dim=-1)                                # [2, m, l, 2]
        lo = (q[0] - self.MARGIN_PX).clamp(-self.CLIP_PX, self.CLIP_PX)
        hi = (q[1] + self.MARGIN_PX).clamp(-self.CLIP_PX, self.CLIP_PX)
        env = torch.stack([lo[..., 1], hi[..., 1], lo[..., 0], hi[..., 0]], -1)   # oy_lo, oy_hi, ox_lo, ox_hi
        return torch.cat([far, env.reshape(-1)])

    def observe(self, locations: torch.Tensor, shapes_host):
        """locations [N, S, M, L, P, 2] normalised (x, y) of an encoder call (the queries are the
        pyramid's own pixels, in pyramid order); shapes_host: numpy int64 [L, 2] (H, W)."""
        self.calls += 1
        if self._pending is not None or (self.calls - 1) % self.every:
            return
        with torch.no_grad():
            res = self.measure(locations, shapes_host)
            host = self._host if (self._host is not None and self._host.numel() == res.numel()) else None
            if host is None:                     # pinned allocations are slow: one buffer per monitor
                host = self._host = torch.empty(res.numel(), dtype=torch.float32, pin_memory=True)
            host.copy_(res, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        self._pending = (host, ev, self.calls)


def measure_envelope(locations: torch.Tensor, spatial_shapes: torch.Tensor):
    """Synchronous form of OffsetMonitor's measurement (tools, tests): numpy float32 [8, 4, 4] or
    None when the call is not an 8-head 4-level one."""
    import numpy as np
    res = OffsetMonitor().measure(locations, _as_int64(spatial_shapes).cpu().numpy()).cpu()
    return None if res.numel() == 1 else np.ascontiguousarray(res[1:].numpy().reshape(8, 4, 4))


_MONITORS = weakref.WeakKeyDictionary()            # module -> OffsetMonitor (not part of the module:
#                                                    events and pinned buffers do not deep-copy)
ADAPTIVE_ROUTING = __import__("os").environ.get("DATR_MSDA_ADAPTIVE", "1") != "0"


class _ZeroRows(torch.autograd.Function):
    """x.masked_fill(mask[..., None], 0) in place on a tensor nobody else reads (the fresh output of
    value_proj); the gradient gets the same treatment.  csrc/msda_prologue.hip::zero_rows_kernel."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        ctx.mark_dirty(x)
        _zero_rows_(x, mask)
        return x

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        g = g.contiguous()
        if not g.is_cuda or g.dtype != torch.float32:
            return g.masked_fill(mask[..., None], 0.0), None
        # g is this node's own incoming gradient (the buffer MSDA's backward produced, handed over
        # through a reshape): nobody reads it after this node, so it is zeroed where it lies
        _zero_rows_(g, mask)
        return g, None


def _zero_rows_(x, mask):
    m = mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)
    with _native.on_device(x.device):
        rc = _native.lib.datr_zero_rows_f32(x.data_ptr(), m.contiguous().data_ptr(), m.numel(), x.shape[-1],
                                            _native.current_stream_ptr(x.device))
    _native.check(rc, "zero_rows")


def zero_padded_rows(value, mask):
    """`value.masked_fill(mask[..., None], 0.0)` (ms_deform_attn.py:101-102); on the device, for the
    contiguous fp32 output of value_proj, in place and touching only the padded rows."""
    if (value.is_cuda and value.dtype == torch.float32 and value.is_contiguous() and value.shape[-1] % 4 == 0
            and mask.is_cuda and mask.shape == value.shape[:-1] and value._base is None
            and (not value.requires_grad or not value.is_leaf)):
        return _ZeroRows.apply(value, mask)
    return value.masked_fill(mask[..., None], 0.0)


class MSDeformAttn(nn.Module):
    """Same constructor, parameter names (state_dict keys `sampling_offsets`,
    `attention_weights`, `value_proj`, `output_proj`), initialisation and forward signature as
    the reference module (ops/modules/ms_deform_attn.py:31-126)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(
                f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a per-head dimension of 16/32/64 takes the "
                          "row-vectorised gfx950 kernels; other sizes use the generic path.")
        self.im2col_step = 64
        self.d_model, self.n_levels = d_model, n_levels
        self.n_heads, self.n_points = n_heads, n_points
        self.sampling_offsets = FastLinear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = FastLinear(d_model, n_heads * n_levels * n_points)
        self.value_proj = FastLinear(d_model, d_model)
        self.output_proj = FastLinear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        # offsets start as a ring of directions (one per head) scaled by the point index
        nn.init.constant_(self.sampling_offsets.weight, 0.0)
        angle = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack([angle.cos(), angle.sin()], dim=-1)
        ring = ring / ring.abs().max(dim=-1, keepdim=True)[0]
        ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            ring[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(ring.reshape(-1))
        nn.init.constant_(self.attention_weights.weight, 0.0)
        nn.init.constant_(self.attention_weights.bias, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None, value=None, grad_slot=None):
        """`value` / `grad_slot`: this module's value projection computed elsewhere (value_projections)
        and where its gradient is to be left; the reference's signature otherwise."""
        N, Len_q, _ = query.shape
        _, Len_in, _ = input_flatten.shape
        H = self.n_heads
        if value is None:
            value = self.value_proj(input_flatten)
            grad_slot = None
        if input_padding_mask is not None:
            value = zero_padded_rows(value, input_padding_mask)
            grad_slot = None
        value = value.view(N, Len_in, H, self.d_model // H)
        fold_wh = False
        if query.is_cuda and MERGE_QUERY_PROJECTIONS:
            # sampling_offsets and attention_weights read the same query: ONE GEMM with the two
            # weight matrices stacked (N = 384 instead of 256 + 128) forward, and one dgrad / one
            # wgrad GEMM backward; the parameters stay separate (state_dict, optimizer)
            w_off, b_off = self.sampling_offsets.weight, self.sampling_offsets.bias
            inv = None
            if reference_points.shape[-1] == 2:
                # 2-d reference points (encoder): offsets are divided by (W_l, H_l) per level.
                # Scaling the 256 rows of the small weight matrix instead folds that division --
                # and its backward -- into the GEMM: two passes over the [N, Lq, 256] offsets
                # less per layer and direction.  (q W) s == q (W s) up to fp32 rounding.
                inv = _inverse_wh(input_spatial_shapes, H, self.n_points)
                fold_wh = True
            w, b = stack_linear(w_off, b_off, self.attention_weights.weight, self.attention_weights.bias, inv)
            both = fast_linear(query, w, b)
            if FUSED_PROLOGUE and (H, self.n_levels, self.n_points) == (8, 4, 4) \
                    and both.dtype == torch.float32 and not reference_points.requires_grad \
                    and reference_points.shape[-1] in (2, 4) \
                    and (fold_wh or reference_points.shape[-1] == 4) and value.dtype == torch.float32:
                route, envelope, mon = 0, None, None
                if fold_wh and ADAPTIVE_ROUTING and Len_q == Len_in:       # encoder self-attention
                    mon = _MONITORS.get(self)
                    if mon is None:
                        mon = _MONITORS[self] = OffsetMonitor()
                    route = mon.poll()
                    envelope = mon.envelope
                if fold_wh and Len_q == Len_in and grad_slot is None and QUERY_GRAD_BACKWARD:
                    # encoder self-attention: one node, whose backward leaves the projection's gradient rows
                    out, locations = _PrologueMSDA.apply(both, reference_points.float(), value, input_spatial_shapes,
                                                         input_level_start_index, self.im2col_step, route, envelope)
                    if mon is not None:
                        mon.observe(locations, _host_meta(_as_int64(input_spatial_shapes),
                                                          _as_int64(input_level_start_index))[0])
                    return self.output_proj(out)
                locations, weights = _Prologue.apply(both, reference_points.float())
                if mon is not None:
                    mon.observe(locations, _host_meta(_as_int64(input_spatial_shapes),
                                                      _as_int64(input_level_start_index))[0])
                out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                                 locations, weights, self.im2col_step, route, envelope, grad_slot)
                return self.output_proj(out)
            n_off = self.sampling_offsets.out_features
            off2, wts2 = _SplitLast.apply(both, n_off)
            offsets = off2.view(N, Len_q, H, self.n_levels, self.n_points, 2)
            weights = wts2.reshape(N, Len_q, H, self.n_levels * self.n_points)
        else:
            offsets = self.sampling_offsets(query).view(N, Len_q, H, self.n_levels, self.n_points, 2)
            weights = self.attention_weights(query).view(N, Len_q, H, self.n_levels * self.n_points)
        weights = F.softmax(weights, -1).view(N, Len_q, H, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2 and fold_wh:
            locations = reference_points[:, :, None, :, None, :] + offsets
        elif reference_points.shape[-1] == 2:
            wh = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] \
                + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get "
                             f"{reference_points.shape[-1]} instead.")
        if value.dtype == torch.float16:     # amp: the kernels run in fp32, like the reference
            out = MSDeformAttnFunction.apply(value.float(), input_spatial_shapes,
                                             input_level_start_index, locations.float(),
                                             weights.float(), self.im2col_step).to(torch.float16)
        else:
            out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                             locations, weights, self.im2col_step)
        return self.output_proj(out)
