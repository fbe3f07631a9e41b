"""Micro-benchmark: datr_wgrad_k256_f32 (weight + bias gradient in one pass) against
dy.t().mm(x) (library GEMM with the committed TunableOp selection) + the column-sum kernel.
    python tools/bench_wgrad_k256.py [--m 88892]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import fused, tuning  # noqa: E402
from bench_gemm import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=88892)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    tuning.enable()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(a.m, 256, device=dev)
    dy = torch.randn(a.m, 256, device=dev) * 0.1
    dw, db = fused.wgrad_k256(dy, x)
    exact_w, exact_b = dy.double().t() @ x.double(), dy.double().sum(0)
    lib_w = dy.t().mm(x)
    t = timeit(lambda: fused.wgrad_k256(dy, x), a.iters)
    tw = timeit(lambda: dy.t().mm(x), a.iters)
    tb = timeit(lambda: fused.column_sums(dy), a.iters)
    gf = 2.0 * a.m * 256 * 256 / 1e9
    print(json.dumps({"M": a.m, "us": round(t, 1), "tflops": round(gf / t, 1), "torch_mm_us": round(tw, 1),
                      "torch_mm_tflops": round(gf / tw, 1), "colsum_us": round(tb, 1),
                      "rel_err_w": ((dw.double() - exact_w).abs().max() / exact_w.abs().max()).item(),
                      "rel_err_w_torch": ((lib_w.double() - exact_w).abs().max() / exact_w.abs().max()).item(),
                      "rel_err_b": ((db.double() - exact_b).abs().max() / exact_b.abs().max()).item()}))


if __name__ == "__main__":
    main()
