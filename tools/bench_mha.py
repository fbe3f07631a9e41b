"""Times the decoder self-attention kernels (csrc/mha_fwd.hip, csrc/mha_bwd.hip) at the training
step's shape (L = 1100, N = 4, 8 heads x 32, DN mask) against PyTorch's memory-efficient SDPA.
Usage: python tools/bench_mha.py [--L 1100] [--N 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd.fused import attention_d32  # noqa: E402


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=1100)
    ap.add_argument("--N", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L, N, H, E = a.L, a.N, 8, 256
    torch.manual_seed(0)
    qk = torch.randn(L, N, 2 * E, device=dev, requires_grad=True)
    v = torch.randn(L, N, E, device=dev, requires_grad=True)
    mask = torch.zeros(L, L, device=dev)
    mask[200:, :200] = float("-inf")
    go = torch.randn(L, N, E, device=dev)
    q, k = qk.split(E, dim=-1)
    out = attention_d32(q, k, v, mask, H)
    t_f = timed(lambda: attention_d32(q, k, v, mask, H))
    t_b = timed(lambda: torch.autograd.grad(out, (qk, v), go, retain_graph=True))
    flops_f = 2 * 2 * L * L * 32 * N * H
    print(f"own fwd {t_f:.1f} us ({flops_f / t_f / 1e6:.1f} TF/s)  own bwd (2 launches + split backward) "
          f"{t_b:.1f} us ({3.5 * flops_f / t_b / 1e6:.1f} TF/s on 7 products)")
    q4, k4, v4 = (x.detach().reshape(L, N, H, 32).permute(1, 2, 0, 3).contiguous().requires_grad_(True)
                  for x in (q, k, v))
    bias = mask.view(1, 1, L, L).expand(N, H, L, L)
    o4 = torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, bias)
    g4 = go.reshape(L, N, H, 32).permute(1, 2, 0, 3).contiguous()
    t_rf = timed(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, bias))
    t_rb = timed(lambda: torch.autograd.grad(o4, (q4, k4, v4), g4, retain_graph=True))
    print(f"torch SDPA fwd {t_rf:.1f} us  bwd {t_rb:.1f} us")


if __name__ == "__main__":
    main()
