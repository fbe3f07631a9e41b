"""Micro-benchmark: datr_gemm_k256_f32 against torch.addmm / mm (hipBLASLt / rocBLAS with the
committed TunableOp selections) at the encoder projection shapes.
    python tools/bench_gemm_k256.py [--m 88892]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import _native, tuning  # noqa: E402


def run(x, b, ldk, ldn, bias, N):
    y = torch.empty(x.shape[0], N, device=x.device)
    rc = _native.lib.datr_gemm_k256_f32(x.data_ptr(), b.data_ptr(), ldk, ldn,
                                        0 if bias is None else bias.data_ptr(), x.shape[0], N,
                                        y.data_ptr(), _native.current_stream_ptr(x.device))
    _native.check(rc, "gemm_k256")
    return y


def timeit(fn, iters):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=88892)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    tuning.enable()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for N, mode in ((256, "fwd"), (256, "dgrad"), (384, "fwd")):
        x = torch.randn(a.m, 256, device=dev)
        w = torch.randn(N, 256, device=dev) * 0.05 if mode == "fwd" else torch.randn(256, N, device=dev) * 0.05
        bias = torch.randn(N, device=dev) if mode == "fwd" else None
        if mode == "fwd":
            mine = lambda: run(x, w, 1, 256, bias, N)
            ref = lambda: torch.addmm(bias, x, w.t())
            exact = (x.double() @ w.double().t() + bias.double())
        else:
            mine = lambda: run(x, w, N, 1, None, N)
            ref = lambda: x.mm(w)
            exact = x.double() @ w.double()
        y, yr = mine(), ref()
        err, err_ref = (y.double() - exact).abs().max().item(), (yr.double() - exact).abs().max().item()
        t, tr = timeit(mine, a.iters), timeit(ref, a.iters)
        gf = 2.0 * a.m * 256 * N / 1e9
        print(json.dumps({"M": a.m, "N": N, "mode": mode, "us": round(t, 1), "tflops": round(gf / t * 1e-3 * 1e3, 1),
                          "torch_us": round(tr, 1), "torch_tflops": round(gf / tr, 1),
                          "max_err_vs_f64": err, "torch_max_err_vs_f64": err_ref}))


if __name__ == "__main__":
    main()
