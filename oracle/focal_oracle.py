"""oracle/focal_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Checkers for the fused focal-loss kernel: the C restatement (oracle/focal_ref.c) and the
reference's own torch formulation restated (/root/reference/models/dino/utils.py:79-104), the
latter differentiable so autograd through it is the backward oracle.  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this."""
import ctypes

import torch
import torch.nn.functional as F

from . import msda_oracle


def focal_sums_c(logits: torch.Tensor, target: torch.Tensor, alpha: float, gamma: float):
    """logits [G,R,C] fp32, target [G,R] int64 -> float64 [G] sums (C loops)."""
    lib = msda_oracle.lib()
    fn = lib.datr_oracle_focal_forward_f32
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int64] * 3 + \
        [ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    lg = logits.detach().cpu().float().contiguous()
    tg = target.detach().cpu().to(torch.int64).contiguous()
    G, R, C = lg.shape
    out = torch.empty(G, dtype=torch.float64)
    assert fn(lg.data_ptr(), tg.data_ptr(), G, R, C, alpha, gamma, out.data_ptr()) == 0
    return out


def focal_sums_torch(logits: torch.Tensor, target: torch.Tensor, alpha: float = 0.25,
                     gamma: float = 2.0):
    """Same quantity with the reference's op sequence (sigmoid, BCE-with-logits, p_t, pow,
    alpha_t) on a one-hot built from the class index; returns [G] sums, differentiable."""
    G, R, C = logits.shape
    onehot = torch.zeros(G, R, C + 1, dtype=logits.dtype, device=logits.device)
    idx = target.clamp(min=0, max=C).where((target >= 0) & (target < C), torch.full_like(target, C))
    onehot.scatter_(2, idx.unsqueeze(-1), 1)
    t = onehot[..., :C]
    prob = logits.sigmoid()
    ce = F.binary_cross_entropy_with_logits(logits, t, reduction="none")
    p_t = prob * t + (1 - prob) * (1 - t)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return loss.sum(dim=(1, 2))
