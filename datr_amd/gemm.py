"""Python front-end of the exact-fp32 MFMA GEMM family with fused epilogues (csrc/gemm_f32.hip).

    gemm_nt(x, w, ...)   y  = epi(x @ w.T)     x [M, K], w [N, K]   -- a linear layer / 1x1 convolution
    gemm_nn(dy, w, ...)  dx = epi(dy @ w)      dy [M, K], w [K, N]  -- its data gradient
    gemm_tn(dy, x, ...)  dw = dy.T @ x         dy [P, M], x [P, N]  -- its weight gradient (split-K)

Operands are 2-d float32 device tensors whose LAST stride is 1 (row slices of wider matrices are
fine: the leading dimension is the row stride).  There is no fallback: an unsupported shape raises.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native

NT, NN, TN = 0, 1, 2
_workspaces = {}

# Who runs the LARGE dense products of the linears / FFN / 1x1 convolutions (>= BIG_ROWS rows: the encoder's
# 88 892-token GEMMs, the wide feature maps) that are hipBLASLt's by default:
#   "library"  hipBLASLt with the per-shape selections of datr_amd/tuning (127-145 TF/s on those shapes), or its
#              default heuristic when the caller asked for that (bench.py --no-tuned-gemm);
#   "own"      this family (129-132 TF/s) -- chosen automatically by `tuning.enable()` when the selections file
#              does not validate against the installed torch / hipBLASLt: the default heuristic then picks
#              ~83 TF/s kernels for the FFN shapes (+12 ms per training step), the own kernels do not depend on
#              any tuning artefact.  DATR_GEMM_BACKEND=own / library forces either side.
BACKEND = "library"
BACKEND_REASON = "default"
BIG_ROWS = 16384
# Row counts (>= BIG_ROWS) for which the selections file holds entries, or None = do not look.  With the "library"
# backend a large product whose row count is NOT among them goes to the own family: the selections are exact-shape
# entries, a training run with multi-scale resizing (or the real C2F size, 87 040 tokens) sees other row counts every
# batch, and there hipBLASLt's default heuristic runs the FFN products at 88-108 TF/s where the own kernels hold
# ~125 whatever the shape (tools/probes and profiles/HISTORY.md, round 5).  DATR_GEMM_UNTUNED=library keeps them on
# the library.
TUNED_ROWS = None


def set_backend(name: str, reason: str, tuned_rows=None) -> None:
    global BACKEND, BACKEND_REASON, TUNED_ROWS
    assert name in ("library", "own")
    BACKEND, BACKEND_REASON = name, reason
    TUNED_ROWS = None if tuned_rows is None else frozenset(tuned_rows)


def within_kernel_limits(shapes, row_strides) -> bool:
    """The size limits of `datr_gemm_f32` (csrc/gemm_f32.hip: 32-bit buffer offsets) over the operands of one
    product, whatever its form -- conservative: an operand must stay below 2 GiB, the output (at most
    rows x the widest dimension in sight, which also bounds rows x ldc / ldr / ldg for row-major outputs of that
    width) below 2^29 elements, and a weight's output-feature count must be a multiple of 4.  `shapes` /
    `row_strides`: per operand (rows, cols) and the row stride in elements; the first operand carries the tokens."""
    rows = shapes[0][0]
    widest = max(max(r if r != rows else 0, c) for r, c in shapes)
    if rows * max(widest, 1) >= 1 << 29:
        return False
    for (r, c), ld in zip(shapes, row_strides):
        if ((r - 1) * ld + c) * 4 >= 0x80000000 - 16:
            return False
        if r * ld >= 1 << 29:
            return False
        if r != rows and r % 4:          # NT: output features; NN: K (already a multiple of 32 by the caller's rule)
            return False
    return True


def own_big(*mats) -> bool:
    """True when the large product over these row-major operands goes to the own family: backend "own" (or a row
    count the library's selections do not cover, see TUNED_ROWS), at least BIG_ROWS rows in the first operand, every
    operand 2-d float32 on the device with a unit last stride, 16-byte aligned rows and a feature count that is a
    multiple of 32 (what every kernel form accepts), and the product inside the kernels' size limits
    (`within_kernel_limits`: beyond them -- e.g. an FFN hidden of >= 262 144 rows -- the library runs it, as before)."""
    rows = mats[0].shape[0]
    if rows < BIG_ROWS:
        return False
    if BACKEND != "own" and (TUNED_ROWS is None or rows in TUNED_ROWS):
        return False
    for t in mats:
        if not (t.dim() == 2 and t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1 and t.shape[1] % 32 == 0
                and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
            return False
    return within_kernel_limits([tuple(t.shape) for t in mats], [t.stride(0) for t in mats])


def _workspace(device, stream: int, floats: int):
    """Scratch per (device, stream), grown on demand; the kernels of one stream run in order, so a
    later call may overwrite what an earlier one has finished with."""
    if floats <= 0:
        return None
    key = (device, stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(int(floats * 1.25) + 1024, device=device, dtype=torch.float32)
        _workspaces[key] = ws
    return ws


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32 and t.is_cuda, (t.shape, t.stride(), t.dtype)
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _run(form, A, B, M, N, K, out, scale, shift, residual, gate, relu, colsum, rowsum_a=False):
    dev = A.device
    if out is None:
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
    epi = _native.GemmEpilogue()
    epi.scale = 0 if scale is None else scale.data_ptr()
    epi.shift = 0 if shift is None else shift.data_ptr()
    epi.residual = 0 if residual is None else residual.data_ptr()
    epi.ldr = 0 if residual is None else _ld(residual)
    epi.gate = 0 if gate is None else gate.data_ptr()
    epi.ldg = 0 if gate is None else _ld(gate)
    epi.relu = int(bool(relu))
    cs = None
    if colsum:
        cs = torch.empty(N, device=dev, dtype=torch.float32)
    epi.colsum = 0 if cs is None else cs.data_ptr()
    ra = torch.empty(M, device=dev, dtype=torch.float32) if rowsum_a else None
    epi.rowsum_a = 0 if ra is None else ra.data_ptr()
    for t in (scale, shift):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
    if residual is not None:
        assert residual.shape == (M, N)
    if gate is not None:
        assert gate.shape == (M, N)
    stream = _native.current_stream_ptr(dev)
    need = int(_native.lib.datr_gemm_workspace_floats(form, M, N, K, int(bool(colsum))))
    if form == TN and scale is not None:
        need = max(need, M * N)
    ws = _workspace(dev, stream, need)
    with _native.on_device(dev):
        rc = _native.lib.datr_gemm_f32(form, A.data_ptr(), _ld(A), B.data_ptr(), _ld(B), M, N, K,
                                       ctypes.addressof(epi), out.data_ptr(), _ld(out),
                                       0 if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), stream)
    _native.check(rc, "gemm_f32")
    if rowsum_a:
        return out, ra
    return (out, cs) if colsum else out


def gemm_nt(x, w, *, out=None, scale=None, shift=None, residual=None, gate=None, relu=False, colsum=False):
    """epi(x @ w.T): x [M, K], w [N, K]; epi = gate(relu(v * scale[n] + shift[n] + residual))."""
    M, K = x.shape
    N, K2 = w.shape
    assert K == K2
    return _run(NT, x, w, M, N, K, out, scale, shift, residual, gate, relu, colsum)


def gemm_nn(dy, w, *, out=None, scale=None, shift=None, residual=None, gate=None, relu=False, colsum=False):
    """epi(dy @ w): dy [M, K], w [K, N]."""
    M, K = dy.shape
    K2, N = w.shape
    assert K == K2
    return _run(NN, dy, w, M, N, K, out, scale, shift, residual, gate, relu, colsum)


def gemm_tn(dy, x, *, out=None, rowscale=None, bias_grad=False):
    """dy.T @ x (optionally * rowscale[:, None]): dy [P, M], x [P, N] -> [M, N]; deterministic split over P.
    bias_grad: also return dy.sum(0) ([M]) -- it falls out of the A fragments, no extra pass over dy."""
    P, M = dy.shape
    P2, N = x.shape
    assert P == P2
    return _run(TN, dy, x, M, N, P, out, rowscale, None, None, None, False, False, rowsum_a=bias_grad)
