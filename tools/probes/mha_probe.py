"""Own MFMA attention forward (csrc/mha_fwd.hip) against PyTorch's SDPA at the decoder
self-attention shape: values, log-sum-exp, time; and whether PyTorch's memory-efficient BACKWARD
accepts the own forward's (out, lse)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import _native  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
L, N, H, D = 1100, 4, 8, 32
E = H * D
qk = torch.randn(L, N, 2 * E, device=dev)
vv = torch.randn(L, N, E, device=dev)
mask = torch.zeros(L, L, dtype=torch.bool, device=dev)
mask[200:, :200] = True
for g in range(10):
    mask[g * 20:(g + 1) * 20, :g * 20] = True
    mask[g * 20:(g + 1) * 20, (g + 1) * 20:200] = True
fmask = torch.zeros(L, L, device=dev).masked_fill(mask, float("-inf"))
q, k = qk.split(E, -1)
q4, k4, v4 = (x.reshape(L, N * H, D).transpose(0, 1).reshape(N, H, L, D) for x in (q, k, vv))


def own():
    out = torch.empty(L, N, E, device=dev)
    lse = torch.empty(N, H, L, device=dev)
    strides = (ctypes.c_int64 * 8)(N * 2 * E, 2 * E, N * 2 * E, 2 * E, N * E, E, N * E, E)
    rc = _native.lib.datr_mha_forward_d32_f32(q.data_ptr(), k.data_ptr(), vv.data_ptr(), fmask.data_ptr(),
                                              L, N, H, ctypes.addressof(strides), D ** -0.5, out.data_ptr(),
                                              lse.data_ptr(), _native.current_stream_ptr(dev))
    _native.check(rc, "mha_fwd")
    return out, lse


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


o_own, lse_own = own()
ref = F.scaled_dot_product_attention(q4, k4, v4, fmask.view(1, 1, L, L))                 # [N,H,L,D]
exact = torch.softmax((q4.double() @ k4.double().transpose(-1, -2)) * D ** -0.5 + fmask.double(), -1) @ v4.double()
o4 = o_own.view(L, N, H, D).permute(1, 2, 0, 3)
print("max |own - f64|", (o4.double() - exact).abs().max().item(), " max |torch - f64|",
      (ref.double() - exact).abs().max().item())
lse_exact = torch.logsumexp((q4.double() @ k4.double().transpose(-1, -2)) * D ** -0.5 + fmask.double(), -1)
print("max |lse - f64|", (lse_own.double() - lse_exact).abs().max().item())
print("own us", round(bench(own), 1), " torch sdpa fwd us",
      round(bench(lambda: F.scaled_dot_product_attention(q4, k4, v4, fmask.view(1, 1, L, L))), 1))

# PyTorch's op: forward outputs and a backward fed with the own (out, lse)
bias = fmask.view(1, 1, L, L).expand(N, H, L, L)
res = torch.ops.aten._scaled_dot_product_efficient_attention(q4, k4, v4, bias, True, 0.0, False)
print("torch lse shape", tuple(res[1].shape), "max |torch lse - own|",
      (res[1][..., :L] - lse_own).abs().max().item())
go = torch.randn(N, H, L, D, device=dev)
gi_ref = torch.ops.aten._scaled_dot_product_efficient_attention_backward(
    go, q4, k4, v4, bias, res[0], res[1], res[2], res[3], 0.0, [True, True, True, False], False)
lse_in = res[1].clone()
lse_in[..., :L] = lse_own
gi_own = torch.ops.aten._scaled_dot_product_efficient_attention_backward(
    go, q4, k4, v4, bias, o4.contiguous(), lse_in, res[2], res[3], 0.0, [True, True, True, False], False)
for a, b, name in zip(gi_ref[:3], gi_own[:3], "qkv"):
    print("d" + name, "max diff", (a - b).abs().max().item(), "scale", a.abs().max().item())
