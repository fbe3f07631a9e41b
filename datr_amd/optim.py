"""AdamW + gradient clipping of the training loop as multi-tensor launches of the own kernels
(csrc/adamw.hip) -- the counterpart of
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm);  optimizer.step()
with `torch.optim.AdamW` (/root/reference/engine.py:99-104, /root/reference/main.py:165).

`FusedClipAdamW` IS a `torch.optim.Optimizer` with AdamW's parameter groups and AdamW's state
(`step`, `exp_avg`, `exp_avg_sq` per parameter: state_dicts load both ways), so schedulers,
checkpoints and `optimizer.param_groups[0]["lr"]` work as with the stock class.  Its `step()` is
AdamW's; `clip_and_step(max_norm, used=...)` is the whole tail of a training step:
  * the global gradient norm by one deterministic two-launch reduction;
  * the update kernel multiplies every gradient by the clip coefficient -- a DEVICE scalar -- as it
    reads it: no pass that rewrites the gradients, no host synchronisation;
  * `used` (int32 device flags, one per parameter in `all_params` order; from
    `datr_amd.dist.GradAllReducer.used_flags()`): a parameter whose flag is 0 is skipped
    entirely, as AdamW skips a parameter whose .grad is None -- the semantics of the reference's
    DistributedDataParallel(find_unused_parameters=True) (main.py:156) for a globally unused parameter.
The tables of the kernels (pointers, lr, weight decay per tensor; the work list) live on the device
and are rebuilt only when a pointer or a hyper-parameter changed (with the flat-bucket reducer the
gradient views are stable, so in steady state a step uploads nothing).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native

_TENSOR_DT = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("step", "<u8"),
                       ("numel", "<i8"), ("lr", "<f4"), ("weight_decay", "<f4"), ("used_index", "<i4"), ("pad", "<i4")])


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Same element order in memory (strides of size-1 dimensions do not matter)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) \
        or t.permute(*sorted(range(t.dim()), key=lambda d: -t.stride(d))).is_contiguous()


class FusedClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        b = {tuple(g["betas"]) + (g["eps"],) for g in self.param_groups}
        if len(b) != 1:
            raise ValueError("FusedClipAdamW: betas / eps must be the same in every parameter group")
        self._tables = None
        self._static = {}           # id(param) -> addresses / strides validated once (_static_of)
        self._index = None          # id(param) -> position in all_params order (the `used` flags' order)
        self.last_norm = None       # device tensor [2]: gradient norm, clip coefficient of the last clip_and_step

    # -- state -------------------------------------------------------------------------------------
    def _state_of(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not (torch.is_tensor(st["step"]) and st["step"].device == p.device and st["step"].dtype == torch.float32):
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device)   # loaded state_dict
        return st

    def set_used_order(self, params):
        """The parameter order the `used` flags of clip_and_step refer to."""
        self._index = {id(p): i for i, p in enumerate(params)}
        self._tables = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = None                     # the moments are other tensors now
        self._static = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = None
        self._static = {}

    def _static_of(self, p):
        """What does not change from step to step for a parameter, validated ONCE (the layout checks of every
        parameter, every step, were 3.6 ms of host time): addresses of the parameter and its moments, element count,
        strides.  Dropped by load_state_dict / add_param_group / a parameter that moved."""
        hit = self._static.get(id(p))
        if hit is not None and hit[0] == p.data_ptr():
            return hit
        if not (p.is_cuda and p.dtype == torch.float32):
            raise TypeError("FusedClipAdamW: float32 device parameters only")
        if not _dense(p):
            raise ValueError("FusedClipAdamW: parameters must be dense")
        st = self._state_of(p)
        for k in ("exp_avg", "exp_avg_sq"):       # moments of a loaded state_dict may carry another layout
            if not _same_layout(st[k], p):
                st[k] = torch.empty_like(p).copy_(st[k])
        hit = self._static[id(p)] = (p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                     st["step"].data_ptr(), p.numel(), p.stride(), p.shape)
        return hit

    def _build(self):
        """(tables, key): device tables over every parameter that has a gradient, in group order."""
        piece = int(_native.lib.datr_adamw_piece_elements())
        rows, pieces, dev, params = [], [], None, []
        index = self._index
        for g in self.param_groups:
            lr, wd = g["lr"], g["weight_decay"]
            for p in g["params"]:
                gr = p.grad
                if gr is None:
                    continue
                pptr, m1, m2, stp, numel, strides, shape = self._static_of(p)
                if gr.dtype != torch.float32:
                    raise TypeError("FusedClipAdamW: float32 device parameters only")
                if gr.stride() != strides and not _same_layout(gr, p):   # e.g. an NCHW gradient of a channels_last weight
                    gr = p.grad = torch.empty_like(p).copy_(gr)
                    self._relaid = True
                dev = p.device
                # -1 = no used flag for this tensor: the kernel then always updates it -- what a plain optimizer does, and
                # the right answer for a parameter the reducer's order does not know (added by add_param_group after
                # set_used_order, or an optimizer over a superset of the reducer's parameters); it never borrows
                # another parameter's flag
                ui = -1 if index is None else index.get(id(p), -1)
                rows.append((pptr, gr.data_ptr(), m1, m2, stp, numel, lr, wd, ui, 0))
                n = len(rows) - 1
                if numel <= piece:
                    pieces.append((n, 0))
                else:
                    pieces += [(n, off) for off in range(0, numel, piece)]
                params.append(p)
        return rows, pieces, dev, params

    def _signature(self):
        """What the device tables depend on, cheaply: learning rate / weight decay of every group and the addresses of
        every parameter and gradient.  Unchanged (the flat-bucket reducer's views, or the caching allocator handing
        the same blocks out every step): the tables of the last step are reused and the per-parameter checks of
        _build -- 4 ms of host time per step, on the critical path where the host is the bound (small images,
        several ranks per host) -- are skipped."""
        sig = []
        for g in self.param_groups:
            sig.append((g["lr"], g["weight_decay"], g["betas"], g["eps"]))
            for p in g["params"]:
                gr = p.grad
                sig.append((p.data_ptr(), 0 if gr is None else gr.data_ptr()))
        return tuple(sig)

    def _tables_for_step(self):
        sig = self._signature()
        if self._tables is not None and self._tables.get("sig") == sig:
            return self._tables
        self._relaid = False
        rows, pieces, dev, params = self._build()
        if not rows:
            return None
        if self._relaid:
            sig = self._signature()                   # (_build re-laid a gradient out: another address)
        key = tuple(rows)
        t = self._tables
        if t is None or t["key"] != key:
            arr = np.array(rows, dtype=_TENSOR_DT)

            def upload(host):
                # through pinned memory, asynchronously: a pageable source makes the copy a host synchronisation,
                # and without the flat-bucket reducer the gradient pointers (hence the tables) change every step.
                # (torch's pinned-memory allocator does not hand a freed block out again before the copies that read
                # it have run: it records the stream of every non_blocking copy)
                pin = torch.empty(host.shape, dtype=host.dtype, pin_memory=True)
                pin.copy_(host)
                return pin, pin.to(dev, non_blocking=True)
            pin_t, dev_t = upload(torch.from_numpy(arr.view(np.uint8).reshape(-1)))
            pin_p, dev_p = upload(torch.tensor(pieces, dtype=torch.int64).reshape(-1, 2))
            t = self._tables = {
                "key": key, "n": len(rows), "npieces": len(pieces), "device": dev,
                "tensors": dev_t, "pieces": dev_p, "pinned": (pin_t, pin_p),
                "partial": torch.empty(len(pieces), dtype=torch.float32, device=dev),
                "norm_coef": torch.empty(2, dtype=torch.float32, device=dev),
            }
        t["params"] = params
        t["sig"] = sig
        return t

    # -- steps -------------------------------------------------------------------------------------
    @torch.no_grad()
    def clip_and_step(self, max_norm: float = 0.0, used: torch.Tensor = None):
        """clip_grad_norm_(max_norm) over this optimizer's parameters (0 = no clipping) + the AdamW step.
        Returns the device tensor [norm, coefficient] (None without clipping)."""
        t = self._tables_for_step()
        if t is None:
            return None
        g0 = self.param_groups[0]
        stream = _native.current_stream_ptr(t["device"])
        if used is not None:
            assert self._index is not None, "set_used_order() first"
            assert used.dtype == torch.int32 and used.is_cuda and used.is_contiguous()
        with _native.on_device(t["device"]):
            coef = 0
            if max_norm and max_norm > 0:
                rc = _native.lib.datr_grad_norm_clip_coef_f32(t["tensors"].data_ptr(), t["pieces"].data_ptr(), t["npieces"],
                                                              float(max_norm), t["partial"].data_ptr(),
                                                              t["norm_coef"].data_ptr(), stream)
                _native.check(rc, "grad_norm_clip_coef")
                coef = t["norm_coef"].data_ptr() + 4
            rc = _native.lib.datr_adamw_step_f32(t["tensors"].data_ptr(), t["n"], t["pieces"].data_ptr(), t["npieces"], coef,
                                                 0 if used is None else used.data_ptr(), g0["betas"][0], g0["betas"][1],
                                                 g0["eps"], stream)
            _native.check(rc, "adamw_step")
        # The kernel writes the parameters through raw pointers: autograd's version counters must move as they would
        # under torch.optim's in-place ops, or caches keyed on (data_ptr, _version) -- the folded frozen-BN weights of
        # pointwise.fold_frozen_bn that an eval-mode forward (engine.evaluate between epochs) reads -- keep serving
        # the weights from before the step.
        torch._C._increment_version(t["params"])
        self.last_norm = t["norm_coef"] if coef else None
        return self.last_norm

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.clip_and_step(0.0)
        return loss
