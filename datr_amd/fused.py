"""Small fused element-wise ops backed by libdatr_hip.so."""
from __future__ import annotations

import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native

# The kernels behind these nodes read raw float32 pointers.  Under torch.autocast (engine.py runs the
# model under it when args.amp is set, /root/reference/engine.py:59) a torch.mm / addmm INSIDE a
# forward would come back in half precision; custom_fwd runs the node with autocast off (its float
# inputs arrive as float32), custom_bwd gives the backward the same state.
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


class _AffineAct(Function):
    """y = act(x * scale[c] + shift[c] (+ res)) over NCHW tensors; one HBM pass each way."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, scale, shift, res, relu, twice=False):
        N, C, H, W = x.shape
        # NHWC (channels_last) tensors are processed in place of layout: channel = i % C
        nhwc = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        fmt = torch.channels_last if nhwc else torch.contiguous_format
        x = x.contiguous(memory_format=fmt)
        y = torch.empty_like(x, memory_format=fmt)
        if res is not None:
            res = res.contiguous(memory_format=fmt)
        inner = 1 if nhwc else H * W
        with _native.on_device(x.device):
            rc = _native.lib.datr_affine_act_forward_f32(
                x.data_ptr(), 0 if res is None else res.data_ptr(), scale.data_ptr(),
                shift.data_ptr(), x.numel(), C, inner, int(relu), y.data_ptr(),
                _native.current_stream_ptr(x.device))
        _native.check(rc, "affine_act_forward")
        ctx.relu, ctx.has_res = bool(relu), res is not None
        ctx.save_for_backward(y if relu else None, scale)
        ctx.shape = (C, inner, fmt)
        ctx.set_materialize_grads(False)         # an unused handle of `twice` arrives as None
        # twice: two handles on y for its two consumers (the next bottleneck's conv1 and identity
        # branch); their gradients are summed inside the backward kernel
        return (y, y.view_as(y)) if twice else y

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy, dy2=None):
        y, scale = ctx.saved_tensors
        C, inner, fmt = ctx.shape
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            return None, None, None, None, None, None
        dy = dy.contiguous(memory_format=fmt)
        if dy2 is not None:
            dy2 = dy2.contiguous(memory_format=fmt)
        dx = torch.empty_like(dy, memory_format=fmt)
        dres = (torch.empty_like(dy, memory_format=fmt)
                if ctx.has_res and ctx.needs_input_grad[3] else None)
        with _native.on_device(dy.device):
            if dy2 is None:
                rc = _native.lib.datr_affine_act_backward_f32(
                    dy.data_ptr(), 0 if y is None else y.data_ptr(), scale.data_ptr(), dy.numel(), C,
                    inner, int(ctx.relu), dx.data_ptr(), 0 if dres is None else dres.data_ptr(),
                    _native.current_stream_ptr(dy.device))
            else:
                rc = _native.lib.datr_affine_act_backward2_f32(
                    dy.data_ptr(), dy2.data_ptr(), 0 if y is None else y.data_ptr(), scale.data_ptr(), dy.numel(),
                    C, inner, int(ctx.relu), dx.data_ptr(), 0 if dres is None else dres.data_ptr(),
                    _native.current_stream_ptr(dy.device))
        _native.check(rc, "affine_act_backward")
        return dx, None, None, dres, None, None


def frozen_bn_act(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                  residual: torch.Tensor = None, relu: bool = True, twice: bool = False):
    """Frozen batch-norm as a per-channel affine, optional residual add, optional ReLU.
    Device float32 NCHW tensors take the fused HIP kernel; anything else evaluates the
    reference's own formula (backbone.py:62-72) op by op.
    twice: return (y, y') -- two handles on the result for its two consumers, whose gradients the backward
    kernel adds on the fly (only with the fused kernel and a gradient to compute; else (y, y))."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
        if twice and FAN_OUT and torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad)):
            return _AffineAct.apply(x, scale.contiguous(), shift.contiguous(), residual, relu, True)
        y = _AffineAct.apply(x, scale.contiguous(), shift.contiguous(), residual, relu)
        return (y, y) if twice else y
    y = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    y = torch.relu(y) if relu else y
    return (y, y) if twice else y


class _GroupNormNHWC(Function):
    """F.group_norm on a channels_last [N, C, H, W] tensor without leaving that layout
    (csrc/groupnorm.hip)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, gamma, beta, groups, eps):
        N, C, H, W = x.shape
        y = torch.empty_like(x, memory_format=torch.channels_last)
        mean = torch.empty(N, groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        part = torch.empty(int(_native.lib.datr_groupnorm_partial_floats(N, H * W, C, groups)), device=x.device,
                           dtype=torch.float32)
        with _native.on_device(x.device):
            rc = _native.lib.datr_groupnorm_nhwc_forward_f32(
                x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), N, H * W, C, groups, eps, y.data_ptr(),
                mean.data_ptr(), rstd.data_ptr(), part.data_ptr(), _native.current_stream_ptr(x.device))
        _native.check(rc, "groupnorm_nhwc_forward")
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.groups = groups
        return y

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        part = torch.empty(int(_native.lib.datr_groupnorm_partial_floats(N, H * W, C, ctx.groups)),
                           device=x.device, dtype=torch.float32)
        with _native.on_device(x.device):
            rc = _native.lib.datr_groupnorm_nhwc_backward_f32(
                dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), N, H * W, C,
                ctx.groups, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), part.data_ptr(),
                _native.current_stream_ptr(x.device))
        _native.check(rc, "groupnorm_nhwc_backward")
        return dx, dgamma, dbeta, None, None


class GroupNormNHWC(torch.nn.GroupNorm):
    """nn.GroupNorm (same parameters, same state_dict names) whose device float32 channels_last
    inputs stay channels_last: one own kernel pair each way instead of ATen's transpose to NCHW and
    back (the input_proj norm of dino.py:111-126 between the NHWC backbone and the [N, HW, C] token
    layout of the transformer).  Other inputs take nn.GroupNorm's path."""

    def forward(self, x):
        C = self.num_channels
        cpg = C // self.num_groups
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and self.affine
                and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
                and C % 4 == 0 and cpg % 4 == 0 and 256 % (C // 4) == 0 and C <= 1024
                and not torch.is_autocast_enabled()):
            return _GroupNormNHWC.apply(x, self.weight, self.bias, self.num_groups, self.eps)
        return super().forward(x)


FFN_FUSED_DZ = os.environ.get("DATR_FFN_FUSED_DZ", "1") != "0"
FFN_FUSED_DZ_MIN_ROWS = int(os.environ.get("DATR_FFN_FUSED_DZ_MIN_ROWS", "16384"))


def _ffn_hidden_gradient(dy2: torch.Tensor, w2: torch.Tensor, h: torch.Tensor):
    """(dz, db1) with dz = (dy2 @ w2) * [h > 0] and db1 = column sums of dz -- the gradient at the FFN's
    hidden pre-activation and linear1's bias gradient (deformable_transformer.py:803-806 under autograd).
    Many rows: ONE launch of the own MFMA GEMM whose epilogue applies the ReLU mask and emits the column
    sums (csrc/gemm_f32.hip) -- no pass over the rows x d_ffn tensor.  Few rows (the decoder): the library
    GEMM followed by the in-place mask + bias-gradient pass (csrc/ffn.hip)."""
    rows = dy2.shape[0]
    cols = w2.shape[1]
    if (FFN_FUSED_DZ and rows >= FFN_FUSED_DZ_MIN_ROWS and w2.is_contiguous() and h.is_contiguous()
            and dy2.shape[1] % 32 == 0 and cols % 4 == 0 and rows * cols < (1 << 29)):
        from . import gemm
        return gemm.gemm_nn(dy2, w2, gate=h, colsum=True)
    dh = dy2.mm(w2)                                       # rows x d_ffn, ours to overwrite
    db1 = torch.empty(cols, device=dh.device, dtype=dh.dtype)
    nblk = int(_native.lib.datr_relu_bwd_bias_partial_rows(rows))
    partial = torch.empty(nblk * cols, device=dh.device, dtype=dh.dtype)
    with _native.on_device(dh.device):
        rc = _native.lib.datr_relu_bwd_bias_f32(dh.data_ptr(), h.data_ptr(), rows, cols, partial.data_ptr(),
                                                db1.data_ptr(), _native.current_stream_ptr(dh.device))
    _native.check(rc, "relu_bwd_bias")
    return dh, db1


FFN_OWN_HIDDEN = os.environ.get("DATR_FFN_OWN_HIDDEN", "0") != "0"


def _ffn_hidden(x2: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor) -> torch.Tensor:
    """h = relu(x2 W1^T + b1): the library GEMM with its bias + ReLU epilogue, or (DATR_FFN_OWN_HIDDEN=1, an A/B
    switch) the own family's NT form with the same epilogue."""
    from . import gemm
    if (FFN_OWN_HIDDEN and x2.shape[0] >= FFN_FUSED_DZ_MIN_ROWS and x2.is_contiguous() and w1.is_contiguous()
            and x2.shape[1] % 32 == 0 and w1.shape[0] % 4 == 0) or gemm.own_big(x2, w1):
        return gemm.gemm_nt(x2, w1, shift=b1.contiguous(), relu=True)
    return torch._addmm_activation(b1, x2, w1.t(), use_gelu=False)


def _linear_fwd(x2: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """x2 W^T + b: the library GEMM, or the own NT form when the large products are the own family's
    (datr_amd.gemm.BACKEND: no valid hipBLASLt selections for this installation)."""
    from . import gemm
    if gemm.own_big(x2, w) and (out is None or gemm.own_big(out)):
        return gemm.gemm_nt(x2, w, shift=b.contiguous(), out=out)
    return torch.addmm(b, x2, w.t()) if out is None else torch.addmm(b, x2, w.t(), out=out)


def _dgrad(dy2: torch.Tensor, w: torch.Tensor, residual: torch.Tensor = None, consume: bool = False) -> torch.Tensor:
    """dy2 W (+ residual): the data gradient of a linear layer, library GEMM or the own NN form.
    consume: `residual` is the caller's own scratch and may hold the result -- the library GEMM then accumulates
    into it (beta = 1, C = D) where `torch.addmm` first copies the residual into a new result (33 us per encoder
    layer for the 91-MB token tensor)."""
    from . import gemm
    if gemm.own_big(dy2, w) and (residual is None or gemm.own_big(residual)):
        return gemm.gemm_nn(dy2, w, residual=residual)
    if residual is None:
        return dy2.mm(w)
    if consume and residual.is_contiguous() and ADDMM_IN_PLACE:
        return residual.addmm_(dy2, w)
    return torch.addmm(residual, dy2, w)


ADDMM_IN_PLACE = os.environ.get("DATR_ADDMM_IN_PLACE", "1") != "0"       # A/B switch


def _wgrad_mm(dy2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """dy2^T x2: library GEMM or the own split-K TN form."""
    from . import gemm
    if gemm.own_big(dy2, x2):
        return gemm.gemm_tn(dy2, x2)
    return dy2.t().mm(x2)


def _ffn_wgrad(dy2: torch.Tensor, x2: torch.Tensor, want_w: bool, want_b: bool):
    """(dy2^T x2, column sums of dy2) of an FFN linear.  Many rows (the encoder, 88 892): the library GEMM
    (124-145 TF/s there) + the column-sum kernel; few rows (the decoder, 4 400): the own split-K kernel, whose
    A fragments give the bias gradient (the library's pick for [2048, 4400] x [4400, 256] runs 141 us in the
    step, the own kernel ~50 us)."""
    if want_w and dy2.shape[0] < FFN_FUSED_DZ_MIN_ROWS and _own_wgrad_applies(dy2, x2):
        from . import gemm
        if want_b:
            return gemm.gemm_tn(dy2, x2, bias_grad=True)
        return gemm.gemm_tn(dy2, x2), None
    from . import gemm
    if want_w and want_b and gemm.own_big(dy2, x2):
        return gemm.gemm_tn(dy2, x2, bias_grad=True)
    dw = _wgrad_mm(dy2, x2) if want_w else None
    db = column_sums(dy2) if want_b else None
    return dw, db


class _FFNRelu(Function):
    """y = linear2(relu(linear1(x))) with hand-placed fusion points:

    forward   h = relu(x W1^T + b1) as ONE hipBLASLt GEMM with a bias+ReLU epilogue
              (`torch._addmm_activation`; bit-identical to linear followed by relu, and no
              separate clamp pass over the rows x d_ffn activation);  y = h W2^T + b2.
    backward  dz = (dy W2) * (h > 0) and db1 = sum_rows(dz) as ONE own MFMA GEMM whose epilogue applies
              the mask and emits the column sums (`_ffn_hidden_gradient`; few rows: library GEMM + one
              in-place pass, csrc/ffn.hip);  dx = dz W1,  dW1 = dz^T x,  dW2 = dy^T h,  db2 = sum_rows(dy).
    Same saved activations as autograd's own graph (x and h)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w1, b1, w2, b2):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        h = _ffn_hidden(x2, w1, b1)
        y = _linear_fwd(h, w2, b2)
        ctx.save_for_backward(x2, h, w1, w2)
        ctx.shape = shape
        return y.view(*shape[:-1], w2.shape[0])

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        x2, h, w1, w2 = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        need = ctx.needs_input_grad
        dw2, db2 = _ffn_wgrad(dy2, h, need[3], need[4])
        dh, db1 = _ffn_hidden_gradient(dy2, w2, h)
        dw1 = _ffn_wgrad(dh, x2, need[1], False)[0]
        dx = _dgrad(dh, w1).view(ctx.shape) if need[0] else None
        return dx, dw1, (db1 if need[2] else None), dw2, db2


def ffn_relu(x: torch.Tensor, linear1: torch.nn.Linear, linear2: torch.nn.Linear) -> torch.Tensor:
    """linear2(relu(linear1(x))).  Device float32 tensors take the fused path above; anything
    else evaluates the reference's own op sequence
    (/root/reference/models/dino/deformable_transformer.py:803-806)."""
    if x.is_cuda and x.dtype == torch.float32 and linear1.bias is not None \
            and linear2.bias is not None and linear1.out_features % 4 == 0:
        return _FFNRelu.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias)
    return linear2(torch.relu(linear1(x)))


class _AddLayerNorm(Function):
    """y = LayerNorm(x + res) (C = 256), forward and backward one HBM pass each
    (csrc/layernorm.hip).  Saves x, res, mean, rstd -- the same bytes autograd's own graph keeps
    (the sum) -- and recomputes x + res in backward."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, res, gamma, beta, eps):
        shape = x.shape
        C = shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        r2 = None
        if res is not None:
            r2 = res.reshape(-1, C)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        with _native.on_device(x.device):
            rc = _native.lib.datr_add_layernorm_forward_f32(
                x2.data_ptr(), 0 if r2 is None else r2.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                rows, C, float(eps), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                _native.current_stream_ptr(x.device))
        _native.check(rc, "add_layernorm_forward")
        ctx.save_for_backward(x2, r2, mean, rstd, gamma)
        ctx.shape = shape
        ctx.res_shape = None if res is None else res.shape
        return y.view(shape)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        x2, r2, mean, rstd, gamma = ctx.saved_tensors
        rows, C = x2.shape
        dy2 = dy.reshape(-1, C)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = torch.empty_like(x2)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        partial = torch.empty(int(_native.lib.datr_add_layernorm_partial_floats(rows)),
                              device=x2.device, dtype=torch.float32)
        with _native.on_device(x2.device):
            rc = _native.lib.datr_add_layernorm_backward_f32(
                dy2.data_ptr(), x2.data_ptr(), 0 if r2 is None else r2.data_ptr(), mean.data_ptr(),
                rstd.data_ptr(), gamma.data_ptr(), rows, C, dx.data_ptr(), partial.data_ptr(),
                dgamma.data_ptr(), dbeta.data_ptr(), _native.current_stream_ptr(x2.device))
        _native.check(rc, "add_layernorm_backward")
        dxv = dx.view(ctx.shape)
        dres = None if r2 is None else dx.view(ctx.res_shape)
        return dxv, dres, dgamma, dbeta, None


class _FFNAddNorm(Function):
    """norm(x + linear2(relu(linear1(x)))): the whole FFN sub-block of a post-norm layer
    (/root/reference/models/dino/deformable_transformer.py:803-806, :879-883) as ONE autograd node --
    the kernels of _FFNRelu and _AddLayerNorm, in the same order.  What the single node buys is the
    backward's fan-in: the gradient of the residual path and the FFN's input gradient meet in the
    beta term of the last GEMM, dx = dsum + dz W1 (`addmm`), instead of a separate add over the
    token tensor (3 x 91 MB per encoder layer)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps, pos=None):
        shape = x.shape
        C = shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        rows = x2.shape[0]
        h = _ffn_hidden(x2, w1, b1)
        y = _linear_fwd(h, w2, b2)
        out = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        ctx.shape = shape
        ctx.query = pos is not None
        if pos is None:
            with _native.on_device(x.device):
                rc = _native.lib.datr_add_layernorm_forward_f32(
                    y.data_ptr(), x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, C, float(eps),
                    out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _native.current_stream_ptr(x.device))
            _native.check(rc, "add_layernorm_forward")
            ctx.save_for_backward(x2, h, w1, w2, y, mean, rstd, gamma)
            return out.view(shape)
        # the next encoder layer's three handles on the output: its query (output + position table, written by
        # the same pass), the value projection's input and the residual; their gradients meet in backward's load
        p2 = pos.reshape(-1, C)
        p2 = p2 if p2.is_contiguous() else p2.contiguous()
        outq = torch.empty_like(x2)
        with _native.on_device(x.device):
            rc = _native.lib.datr_add_layernorm_forward_query_f32(
                y.data_ptr(), x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), p2.data_ptr(), rows, C, float(eps),
                out.data_ptr(), outq.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _native.current_stream_ptr(x.device))
        _native.check(rc, "add_layernorm_forward_query")
        ctx.save_for_backward(x2, h, w1, w2, y, mean, rstd, gamma)
        ctx.pos_shape = pos.shape
        ctx.set_materialize_grads(False)
        out = out.view(shape)
        return outq.view(shape), out.view_as(out), out.view_as(out)

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dout, dv=None, dr=None):
        x2, h, w1, w2, y, mean, rstd, gamma = ctx.saved_tensors
        rows, C = x2.shape
        dpos = None
        if ctx.query:
            # (query, value, residual): the position table's gradient is the query's, as it arrives
            if ctx.needs_input_grad[8] and dout is not None:
                dpos = dout.reshape(ctx.pos_shape)
            live = [g for g in (dout, dv, dr) if g is not None]
            if not live:
                live = [torch.zeros(ctx.shape, device=x2.device, dtype=x2.dtype)]
        else:
            live = [dout]
        live = [g.reshape(-1, C) for g in live]
        live = [g if g.is_contiguous() else g.contiguous() for g in live]
        dsum = torch.empty_like(x2)                       # gradient of (x + ffn(x)): both addends get it
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        db2 = torch.empty_like(gamma)                     # linear2's bias gradient = the column sums of dsum
        partial = torch.empty(int(_native.lib.datr_add_layernorm_partial_floats(rows)),
                              device=x2.device, dtype=torch.float32)
        stream = _native.current_stream_ptr(x2.device)
        with _native.on_device(x2.device):
            # ... which fall out of the LayerNorm backward's own pass over dsum (no column-sum launches)
            if len(live) == 1:
                rc = _native.lib.datr_add_layernorm_backward_colsum_f32(
                    live[0].data_ptr(), y.data_ptr(), x2.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                    rows, C, dsum.data_ptr(), partial.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), db2.data_ptr(), stream)
            else:
                rc = _native.lib.datr_add_layernorm_backward_fanin_f32(
                    live[0].data_ptr(), live[1].data_ptr(), live[2].data_ptr() if len(live) > 2 else 0,
                    y.data_ptr(), x2.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                    rows, C, dsum.data_ptr(), partial.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), db2.data_ptr(), stream)
        _native.check(rc, "add_layernorm_backward")
        need = ctx.needs_input_grad
        dw2 = _ffn_wgrad(dsum, h, need[3], False)[0]
        if not need[4]:
            db2 = None
        dh, db1 = _ffn_hidden_gradient(dsum, w2, h)
        dw1 = _ffn_wgrad(dh, x2, need[1], False)[0]
        dx = _dgrad(dh, w1, residual=dsum, consume=True).view(ctx.shape) if need[0] else None     # dsum's last use
        return (dx, dw1, db1 if need[2] else None, dw2, db2, dgamma if need[5] else None,
                dbeta if need[6] else None, None, dpos)


def ffn_add_norm(x: torch.Tensor, linear1: torch.nn.Linear, linear2: torch.nn.Linear, norm: torch.nn.LayerNorm,
                 next_pos: torch.Tensor = None):
    """norm(x + linear2(relu(linear1(x)))) as one node, or None when the fused kernels do not apply
    (the caller then composes ffn_relu / add_layer_norm or the reference's ops).  With `next_pos` (the position
    table the NEXT encoder layer adds to its query, deformable_transformer.py:789-798) the node returns that layer's
    three handles (output + next_pos, output, output) -- query, value input, residual -- and sums their gradients
    while it loads them, where `q = src + pos` and the three-way gradient sum each cost passes over the tokens."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and linear1.bias is not None
            and linear2.bias is not None and linear1.out_features % 4 == 0 and linear2.out_features == 256
            and norm.elementwise_affine and norm.bias is not None and tuple(norm.normalized_shape) == (256,)
            and torch.is_grad_enabled()):
        return None
    if next_pos is not None:
        if not (next_pos.shape == x.shape and next_pos.dtype == x.dtype and next_pos.device == x.device):
            return None
        return _FFNAddNorm.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm.weight,
                                 norm.bias, norm.eps, next_pos)
    return _FFNAddNorm.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm.weight,
                             norm.bias, norm.eps)


def add_layer_norm(x: torch.Tensor, res: torch.Tensor, norm: torch.nn.LayerNorm) -> torch.Tensor:
    """norm(x + res) -- the tail of every post-norm sub-block
    (/root/reference/models/dino/deformable_transformer.py:796-806, :856-893).  Device float32
    tensors with 256 channels and an affine LayerNorm take the fused kernels; anything else
    evaluates the reference's two ops."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and x.shape == res.shape \
            and norm.elementwise_affine and norm.bias is not None \
            and tuple(norm.normalized_shape) == (256,):
        return _AddLayerNorm.apply(x, res, norm.weight, norm.bias, norm.eps)
    return norm(x + res)


def layer_norm(x: torch.Tensor, norm: torch.nn.LayerNorm) -> torch.Tensor:
    """norm(x) for the two LayerNorms of the path that follow no residual add (`enc_output_norm`
    over all encoder tokens, deformable_transformer.py:335-336, and the decoder's output norm,
    :725) through the same one-pass kernels (csrc/layernorm.hip with a null residual)."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and norm.elementwise_affine \
            and norm.bias is not None and tuple(norm.normalized_shape) == (256,):
        return _AddLayerNorm.apply(x, None, norm.weight, norm.bias, norm.eps)
    return norm(x)


def layer_norm_class_max(x: torch.Tensor, norm: torch.nn.LayerNorm, head: torch.nn.Linear, row_mask=None,
                         row_fill=None) -> torch.Tensor:
    """head(norm(x)).max(-1)[0] for x [..., 256] -- the score the two-stage selection ranks encoder tokens by
    (/root/reference/models/dino/deformable_transformer.py:335-342) -- in one pass (csrc/layernorm.hip:
    neither the normalised rows nor the logits are written).  row_mask (bool, one entry per row) / row_fill [256]:
    masked rows are evaluated as if x[r] were row_fill.  No gradient: callers use it where the reference's
    values feed top-k only.  None when the kernel does not apply."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and norm.elementwise_affine
            and norm.bias is not None and tuple(norm.normalized_shape) == (256,) and isinstance(head, torch.nn.Linear)
            and head.bias is not None and head.in_features == 256 and head.out_features <= 16
            and head.weight.dtype == torch.float32 and not torch.is_autocast_enabled()):
        return None
    x2 = x.reshape(-1, 256)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    out = torch.empty(x2.shape[0], device=x.device, dtype=torch.float32)
    mask = fill = None
    if row_mask is not None:
        mask = row_mask.reshape(-1).contiguous()
        fill = row_fill.detach().contiguous()
        assert mask.dtype == torch.bool and mask.numel() == x2.shape[0] and fill.shape == (256,) and fill.dtype == torch.float32
    with torch.no_grad(), torch.cuda.device(x.device):
        rc = _native.lib.datr_layernorm_class_max_f32(
            x2.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), head.weight.contiguous().data_ptr(),
            head.bias.data_ptr(), x2.shape[0], 256, head.out_features, float(norm.eps),
            0 if mask is None else mask.data_ptr(), 0 if fill is None else fill.data_ptr(), out.data_ptr(),
            _native.current_stream_ptr(x.device))
    _native.check(rc, "layernorm_class_max")
    return out.view(x.shape[:-1])


def topk_rows(scores: torch.Tensor, k: int):
    """(values, indices) of the k largest entries of every row of `scores` [rows, n], sorted by
    descending value, EQUAL values by ascending index (csrc/topk.hip): one total order, so the
    selection does not depend on device or launch -- `torch.topk` leaves the order, and at the
    k-th value the membership, of ties open (its CPU and GPU implementations differ).  With
    distinct scores this IS torch.topk's result.  On the CPU (host-side tests only: the product
    runs on the device) the same order comes from a stable sort."""
    assert scores.dim() == 2 and 1 <= k <= scores.shape[1]
    if not scores.is_cuda:
        key = scores.float()
        key = torch.where(torch.isnan(key), torch.full_like(key, float("inf")), key)
        order = torch.argsort(key, dim=1, descending=True, stable=True)
        nan_first = torch.argsort((~torch.isnan(scores.float())).to(torch.int8).gather(1, order), dim=1, stable=True)
        idx = order.gather(1, nan_first)[:, :k]
        return scores.gather(1, idx), idx
    if k > 1024:
        raise NotImplementedError("topk_rows: k <= 1024 on the device (csrc/topk.hip)")
    x = scores.detach().float().contiguous()
    rows, n = x.shape
    idx = torch.empty(rows, k, dtype=torch.int64, device=x.device)
    val = torch.empty(rows, k, dtype=torch.float32, device=x.device)
    with _native.on_device(x.device):
        rc = _native.lib.datr_topk_rows_f32(x.data_ptr(), rows, n, k, idx.data_ptr(), val.data_ptr(),
                                            _native.current_stream_ptr(x.device))
    _native.check(rc, "topk_rows")
    return scores.gather(1, idx) if scores.requires_grad or scores.dtype != torch.float32 else val, idx


class _RefineBoxes(Function):
    """sigmoid(delta + inverse_sigmoid(ref)) as one launch each way (csrc/refine.hip)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, delta, ref, eps):
        d, r = delta.contiguous(), ref.contiguous()
        out = torch.empty_like(d)
        with _native.on_device(d.device):
            rc = _native.lib.datr_refine_boxes_forward_f32(d.data_ptr(), r.data_ptr(), d.numel(), eps, out.data_ptr(),
                                                           _native.current_stream_ptr(d.device))
        _native.check(rc, "refine_boxes_forward")
        ctx.save_for_backward(out, r)
        ctx.eps = eps
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, g):
        out, r = ctx.saved_tensors
        g = g.contiguous()
        dd = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(out) if ctx.needs_input_grad[1] else None
        if dd is None and dr is None:
            return None, None, None
        with _native.on_device(g.device):
            rc = _native.lib.datr_refine_boxes_backward_f32(g.data_ptr(), out.data_ptr(), r.data_ptr(), g.numel(), ctx.eps,
                                                            0 if dd is None else dd.data_ptr(),
                                                            0 if dr is None else dr.data_ptr(),
                                                            _native.current_stream_ptr(g.device))
        _native.check(rc, "refine_boxes_backward")
        return dd, dr, None


def refine_boxes(delta: torch.Tensor, ref: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """sigmoid(delta + inverse_sigmoid(ref)): the decoder's iterative box refinement
    (/root/reference/models/dino/deformable_transformer.py:738-744, dino.py:316-322).  Device float32 tensors
    of one shape take the fused kernel; anything else evaluates the reference's op sequence."""
    if delta.is_cuda and delta.dtype == torch.float32 and ref.dtype == torch.float32 and delta.shape == ref.shape:
        return _RefineBoxes.apply(delta, ref, eps)
    from .nested import inverse_sigmoid
    return (delta + inverse_sigmoid(ref, eps)).sigmoid()


FAN_OUT = os.environ.get("DATR_FAN_OUT", "1") != "0"


class _FanOut(torch.autograd.Function):
    """n aliases of x for n consumers; the backward adds their gradients in ONE pass (csrc/addn.hip)
    where autograd would add them pairwise (n - 1 launches of 3 passes each)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        ctx.set_materialize_grads(False)         # an unused handle arrives as None, not as a zero tensor
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [g for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        ref = live[0]
        if not (ref.is_cuda and ref.dtype == torch.float32 and 2 <= len(live) <= 8
                and all(g.shape == ref.shape and g.dtype == ref.dtype for g in live)):
            total = live[0]
            for g in live[1:]:
                total = total + g
            return total, None
        # same dense layout for all (e.g. six transposed views of contiguous gradients): sum in storage order
        out = torch.empty_like(ref)
        if out.stride() != ref.stride() or any(g.stride() != ref.stride() for g in live):
            live = [g.contiguous() for g in live]
            ref = live[0]
            out = torch.empty_like(ref)
        ptrs = (ctypes.c_void_p * len(live))(*[g.data_ptr() for g in live])
        with _native.on_device(ref.device):
            rc = _native.lib.datr_add_n_f32(ctypes.addressof(ptrs), len(live), ref.numel(), out.data_ptr(),
                                            _native.current_stream_ptr(ref.device))
        _native.check(rc, "add_n")
        return out, None


def fan_out(x: torch.Tensor, n: int):
    """n handles on x, one per consumer, whose gradients are summed by one kernel; plain references when
    x needs no gradient or is not a float32 device tensor."""
    if not (FAN_OUT and n >= 2 and n <= 8 and x.is_cuda and x.dtype == torch.float32 and x.requires_grad
            and torch.is_grad_enabled()):
        return (x,) * n
    return _FanOut.apply(x, n)


def column_sums(x2: torch.Tensor) -> torch.Tensor:
    """sum over rows of a contiguous [rows, cols] fp32 device matrix (cols % 4 == 0)."""
    rows, cols = x2.shape
    out = torch.empty(cols, device=x2.device, dtype=x2.dtype)
    nblk = int(_native.lib.datr_relu_bwd_bias_partial_rows(rows))
    partial = torch.empty(nblk * cols, device=x2.device, dtype=x2.dtype)
    with _native.on_device(x2.device):
        rc = _native.lib.datr_colsum_f32(x2.data_ptr(), rows, cols, partial.data_ptr(), out.data_ptr(),
                                         _native.current_stream_ptr(x2.device))
    _native.check(rc, "colsum")
    return out


class _LinearFn(Function):
    """y = x W^T + b.  Backward: the data-gradient GEMM autograd would run; weight AND bias gradient from one
    launch of the own split-K kernel (datr_amd.gemm.gemm_tn: deterministic, the bias gradient falls out of
    its A fragments) from 1 024 rows on, below that the library GEMM + the deterministic column-sum kernel
    (csrc/ffn.hip) instead of ATen's generic reduction."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, b):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        # the result is allocated in its final shape (the GEMM writes through a 2-d view of it): the
        # caller gets a tensor that is nobody's view and may hand it to an in-place op (msda._ZeroRows)
        out = torch.empty(*shape[:-1], w.shape[0], device=x.device, dtype=x.dtype)
        _linear_fwd(x2, w, b, out=out.view(-1, w.shape[0]))
        ctx.save_for_backward(x2, w)
        ctx.shape = shape
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        need = ctx.needs_input_grad
        dx = _dgrad(dy2, w).view(ctx.shape) if need[0] else None
        if need[1] and _own_wgrad_applies(dy2, x2):
            # weight gradient as the deterministic split-K product of the own GEMM family; the bias
            # gradient falls out of its A fragments (no column-sum launches)
            from . import gemm
            if need[2]:
                dw, db = gemm.gemm_tn(dy2, x2, bias_grad=True)
            else:
                dw, db = gemm.gemm_tn(dy2, x2), None
            return dx, dw, db
        dw = dy2.t().mm(x2) if need[1] else None
        db = column_sums(dy2) if need[2] else None
        return dx, dw, db


OWN_WGRAD = os.environ.get("DATR_OWN_LINEAR_WGRAD", "1") != "0"
OWN_WGRAD_MIN_ROWS = int(os.environ.get("DATR_OWN_LINEAR_WGRAD_MIN_ROWS", "1024"))


def _own_wgrad_applies(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    """dy2 [rows, out], x2 [rows, in]: rows enough to split, feature counts multiples of 4, row-contiguous."""
    return (OWN_WGRAD and dy2.shape[0] >= OWN_WGRAD_MIN_ROWS and dy2.shape[1] % 4 == 0 and x2.shape[1] % 4 == 0
            and dy2.stride(1) == 1 and x2.stride(1) == 1 and dy2.stride(0) % 4 == 0 and x2.stride(0) % 4 == 0
            and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0)


class _LinearReluFn(Function):
    """relu(x W^T + b): bias + ReLU in the GEMM's epilogue (`torch._addmm_activation`, bit-identical to linear
    followed by relu); backward = the ReLU gate in one pass (csrc/affine_act.hip) + _LinearFn's backward."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, w, b):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        y = torch._addmm_activation(b, x2, w.t(), use_gelu=False)
        ctx.save_for_backward(x2, w, y)
        ctx.shape = shape
        return y.view(*shape[:-1], w.shape[0])

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        cols = w.shape[0]
        dy2 = dy.reshape(-1, cols)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dz = torch.empty_like(dy2)
        from .pointwise import _ones
        with _native.on_device(dy2.device):
            rc = _native.lib.datr_affine_act_backward_f32(dy2.data_ptr(), y.data_ptr(), _ones(cols, dy2.device).data_ptr(),
                                                          dy2.numel(), cols, 1, 1, dz.data_ptr(), 0,
                                                          _native.current_stream_ptr(dy2.device))
        _native.check(rc, "affine_act_backward")
        need = ctx.needs_input_grad
        dx = dz.mm(w).view(ctx.shape) if need[0] else None
        if need[1] and _own_wgrad_applies(dz, x2):
            from . import gemm
            if need[2]:
                dw, db = gemm.gemm_tn(dz, x2, bias_grad=True)
            else:
                dw, db = gemm.gemm_tn(dz, x2), None
            return dx, dw, db
        dw = dz.t().mm(x2) if need[1] else None
        db = column_sums(dz) if need[2] else None
        return dx, dw, db


def linear_relu(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """relu(F.linear(x, weight, bias)); device float32 inputs take _LinearReluFn, anything else the two ops."""
    if x.is_cuda and x.dtype == torch.float32 and bias is not None and weight.shape[0] % 4 == 0 \
            and x.dim() >= 2 and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        return _LinearReluFn.apply(x, weight, bias)
    return torch.relu(torch.nn.functional.linear(x, weight, bias))


NOGRAD_OWN_MIN_ROWS = int(os.environ.get("DATR_NOGRAD_LINEAR_OWN_MIN_ROWS", "16384"))


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """F.linear; device float32 inputs with a bias of a multiple of 4 features take _LinearFn."""
    if x.is_cuda and x.dtype == torch.float32 and bias is not None and weight.shape[0] % 4 == 0 \
            and x.dim() >= 2 and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        return _LinearFn.apply(x, weight, bias)
    if x.is_cuda and x.dtype == torch.float32 and bias is not None and x.dim() >= 2 and x.is_contiguous() \
            and weight.is_contiguous() and x.shape[-1] % 32 == 0 and weight.shape[0] % 4 == 0 \
            and x.numel() // x.shape[-1] >= NOGRAD_OWN_MIN_ROWS and not torch.is_autocast_enabled() \
            and x.data_ptr() % 16 == 0 and x.shape[-1] <= 512:
        # without autograd (the two-stage pass over all 88 892 encoder tokens, deformable_transformer.py:329-336):
        # F.linear on a 3-d input is hipBLASLt's bias-epilogue entry, whose pick for [88 892, 256] x [256, 256] ran
        # 639 us in the round-4 step (profiles/r04_library_gemm_calls.txt) where the own NT form runs 111 us.  Short
        # reductions only (tools/probes/nograd_linear_probe.py: K = 256 own 127 / 167 us against 131 / 185 us for
        # N = 256 / 384, a tie at N = 2048; K = 2048 the library's 668 us against 749 us -- the eval / teacher FFN)
        from . import gemm
        y = gemm.gemm_nt(x.reshape(-1, x.shape[-1]), weight, shift=bias.contiguous())
        return y.view(*x.shape[:-1], weight.shape[0])
    return torch.nn.functional.linear(x, weight, bias)


class FastLinear(torch.nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) whose backward computes the bias
    gradient with the column-sum kernel."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


# ---------------------------------------------------------------------------------------------
# Scaled-dot-product attention, head_dim 32: own exact-fp32 MFMA forward (csrc/mha_fwd.hip) and
# backward (csrc/mha_bwd.hip: query-stationary dQ + key-stationary dK/dV, fed with the forward's
# output and log-sum-exp).
# ---------------------------------------------------------------------------------------------
class _AttentionD32(Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, v, mask, heads):
        L, N, E = q.shape
        out = torch.empty(L, N, E, device=q.device, dtype=torch.float32)
        lse = torch.empty(N, heads, L, device=q.device, dtype=torch.float32)
        strides = (ctypes.c_int64 * 8)(q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                                       v.stride(0), v.stride(1), out.stride(0), out.stride(1))
        with _native.on_device(q.device):
            rc = _native.lib.datr_mha_forward_d32_f32(
                q.data_ptr(), k.data_ptr(), v.data_ptr(), 0 if mask is None else mask.data_ptr(), L, N,
                heads, ctypes.addressof(strides), 32 ** -0.5, out.data_ptr(), lse.data_ptr(),
                _native.current_stream_ptr(q.device))
        _native.check(rc, "mha_forward_d32")
        ctx.save_for_backward(q, k, v, out, lse, mask)
        ctx.heads = heads
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dout):
        q, k, v, out, lse, mask = ctx.saved_tensors
        L, N, E = q.shape
        H = ctx.heads
        if dout.stride(-1) != 1 or dout.stride(0) % 4 or dout.stride(1) % 4:
            dout = dout.contiguous()
        # one buffer for the three gradients: q and k are column slices of ONE merged projection in
        # the decoder (transformer._self_attention), so their gradients land side by side
        grads = torch.empty(L, N, 3 * E, device=q.device, dtype=torch.float32)
        dq, dk, dv = grads[..., :E], grads[..., E:2 * E], grads[..., 2 * E:]
        delta = torch.empty(N, H, L, device=q.device, dtype=torch.float32)
        strides = (ctypes.c_int64 * 16)(q.stride(0), q.stride(1), k.stride(0), k.stride(1),
                                        v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                        dout.stride(0), dout.stride(1), dq.stride(0), dq.stride(1),
                                        dk.stride(0), dk.stride(1), dv.stride(0), dv.stride(1))
        with _native.on_device(q.device):
            rc = _native.lib.datr_mha_backward_d32_f32(
                dout.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                0 if mask is None else mask.data_ptr(), L, N, H, ctypes.addressof(strides), 32 ** -0.5,
                delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                _native.current_stream_ptr(q.device))
        _native.check(rc, "mha_backward_d32")
        return dq, dk, dv, None, None


class _AttentionQKD32(Function):
    """_AttentionD32 for a MERGED query / key projection qk [L, N, 2 E] (q = qk[..., :E], k = qk[..., E:]) and
    v [L, N, E], in any row order the kernels' strides express -- sequence-major or batch-major memory.  The
    output and the gradients are allocated in the memory order of qk; the gradient of qk comes back as one
    tensor (no split-backward concatenation)."""

    @staticmethod
    def _like(ref, width):
        L, N = ref.shape[:2]
        if ref.stride(1) > ref.stride(0):                  # batch-major memory behind the [L, N, .] view
            return torch.empty(N, L, width, device=ref.device, dtype=torch.float32).transpose(0, 1)
        return torch.empty(L, N, width, device=ref.device, dtype=torch.float32)

    @staticmethod
    @_amp_fwd
    def forward(ctx, qk, v, mask, heads):
        L, N, E = v.shape
        out = _AttentionQKD32._like(qk, E)
        lse = torch.empty(N, heads, L, device=qk.device, dtype=torch.float32)
        strides = (ctypes.c_int64 * 8)(qk.stride(0), qk.stride(1), qk.stride(0), qk.stride(1),
                                       v.stride(0), v.stride(1), out.stride(0), out.stride(1))
        with _native.on_device(qk.device):
            rc = _native.lib.datr_mha_forward_d32_f32(
                qk.data_ptr(), qk.data_ptr() + 4 * E, v.data_ptr(), 0 if mask is None else mask.data_ptr(), L, N,
                heads, ctypes.addressof(strides), 32 ** -0.5, out.data_ptr(), lse.data_ptr(),
                _native.current_stream_ptr(qk.device))
        _native.check(rc, "mha_forward_d32")
        ctx.save_for_backward(qk, v, out, lse, mask)
        ctx.heads = heads
        return out

    @staticmethod
    @once_differentiable
    @_amp_bwd
    def backward(ctx, dout):
        qk, v, out, lse, mask = ctx.saved_tensors
        L, N, E = v.shape
        H = ctx.heads
        if dout.stride(-1) != 1 or dout.stride(0) % 4 or dout.stride(1) % 4:
            dout = dout.contiguous()
        dqk = _AttentionQKD32._like(qk, 2 * E)
        dv = _AttentionQKD32._like(qk, E)
        delta = torch.empty(N, H, L, device=qk.device, dtype=torch.float32)
        strides = (ctypes.c_int64 * 16)(qk.stride(0), qk.stride(1), qk.stride(0), qk.stride(1),
                                        v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                        dout.stride(0), dout.stride(1), dqk.stride(0), dqk.stride(1),
                                        dqk.stride(0), dqk.stride(1), dv.stride(0), dv.stride(1))
        with _native.on_device(qk.device):
            rc = _native.lib.datr_mha_backward_d32_f32(
                dout.data_ptr(), qk.data_ptr(), qk.data_ptr() + 4 * E, v.data_ptr(), out.data_ptr(), lse.data_ptr(),
                0 if mask is None else mask.data_ptr(), L, N, H, ctypes.addressof(strides), 32 ** -0.5,
                delta.data_ptr(), dqk.data_ptr(), dqk.data_ptr() + 4 * E, dv.data_ptr(),
                _native.current_stream_ptr(qk.device))
        _native.check(rc, "mha_backward_d32")
        return dqk, dv, None, None


def attention_qk_d32(qk: torch.Tensor, v: torch.Tensor, mask, heads: int) -> torch.Tensor:
    """attention_d32 with the query and key given as ONE tensor qk [L, N, 2 * heads * 32] (the decoder's merged
    projection); the [L, N, .] views may sit on batch-major memory (transposed [N, L, .] tensors).  Returns
    [L, N, heads * 32] in the memory order of qk."""
    E = heads * 32
    assert qk.is_cuda and qk.dtype == v.dtype == torch.float32 and qk.shape == (*v.shape[:2], 2 * E) and v.shape[-1] == E
    assert qk.stride(-1) == v.stride(-1) == 1
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.shape == (qk.shape[0], qk.shape[0]) and mask.is_contiguous()
    return _AttentionQKD32.apply(qk, v, mask, heads)


def attention_d32(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask, heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(32) + mask) v per (batch, head) for sequence-first [L, N, heads * 32]
    device float32 tensors (last dim contiguous, other strides multiples of 4 -- column slices of a
    merged projection are fine); mask: additive float [L, L] or None.  Returns [L, N, heads * 32]
    contiguous, i.e. what nn.MultiheadAttention hands to out_proj."""
    assert q.is_cuda and q.dtype == torch.float32 and q.shape == k.shape == v.shape
    assert q.shape[-1] == heads * 32 and q.stride(-1) == k.stride(-1) == v.stride(-1) == 1
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.shape == (q.shape[0], q.shape[0]) and mask.is_contiguous()
    return _AttentionD32.apply(q, k, v, mask, heads)
