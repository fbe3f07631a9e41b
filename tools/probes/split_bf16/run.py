"""EXPERIMENT (not the product path): fp32 GEMM on the bf16 matrix pipe by exact three-way operand splits
(gemm_split_bf16.hip) -- accuracy against float64 and time against the product's fp32-MFMA GEMM family and the
library, at the encoder FFN shapes.    python tools/probes/split_bf16/run.py"""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
from datr_amd import gemm, tuning  # noqa: E402

tuning.enable()
lib = ctypes.CDLL(os.path.join(HERE, "libsplit_bf16.so"))
lib.split_bf16_gemm_nt.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for M, N, K in [(88892, 2048, 256), (88892, 256, 2048), (88892, 256, 256), (16800, 512, 128)]:
    x = torch.randn(M, K, device=dev) * 2.0 + 0.3
    w = torch.randn(N, K, device=dev) * K ** -0.5
    out = torch.empty(M, N, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    idx = torch.randint(0, M, (256,), device=dev)
    ref = x[idx].double() @ w.double().t()
    scale = (x[idx].double().abs() @ w.double().abs().t())          # sum |a||b|: the natural error scale
    line = [f"M={M} N={N} K={K}"]
    for products in (1, 3, 6):
        run = lambda: lib.split_bf16_gemm_nt(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, products, stream)
        assert run() == 0
        torch.cuda.synchronize()
        err = ((out[idx].double() - ref).abs() / scale).max().item()
        t = timeit(run)
        line.append(f"split x{products}: {t:7.1f} us {2.0 * M * N * K / t * 1e-6:6.1f} TF/s err {err:.2e}")
    y = gemm.gemm_nt(x, w)
    err = ((y[idx].double() - ref).abs() / scale).max().item()
    t = timeit(lambda: gemm.gemm_nt(x, w))
    line.append(f"own fp32 MFMA: {t:7.1f} us {2.0 * M * N * K / t * 1e-6:6.1f} TF/s err {err:.2e}")
    t = timeit(lambda: torch.mm(x, w.t(), out=out))
    err = ((out[idx].double() - ref).abs() / scale).max().item()
    line.append(f"library: {t:7.1f} us {2.0 * M * N * K / t * 1e-6:6.1f} TF/s err {err:.2e}")
    print("\n    ".join(line), flush=True)
