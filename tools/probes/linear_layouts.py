"""Library GEMM time of a linear layer y = x W^T + b by how the weight is handed over: the parameter's own
[N, K] layout (column-major "TN") against a contiguous transposed copy [K, N] ("NN"), with and without the
bias / ReLU epilogue, at the encoder's row count.  TunableOp selections loaded as in the training step.
    python tools/probes/linear_layouts.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import tuning  # noqa: E402

tuning.enable()
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


for M, K, N in [(88892, 256, 2048), (88892, 2048, 256), (88892, 256, 256), (88892, 256, 384), (4400, 256, 2048), (4400, 256, 256)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * K ** -0.5
    wt = w.t().contiguous()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    r = {
        "addmm TN": timeit(lambda: torch.addmm(b, x, w.t(), out=out)),
        "addmm NN": timeit(lambda: torch.addmm(b, x, wt, out=out)),
        "relu TN": timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)),
        "relu NN": timeit(lambda: torch._addmm_activation(b, x, wt, use_gelu=False)),
        "mm TN": timeit(lambda: torch.mm(x, w.t(), out=out)),
        "mm NN": timeit(lambda: torch.mm(x, wt, out=out)),
        "transpose": timeit(lambda: w.t().contiguous()),
    }
    print(f"M={M} K={K} N={N}: " + "  ".join(f"{k} {v:.1f}" for k, v in r.items()), flush=True)
