"""Small weight-gradient problems of the decoder (4 400 rows) on the TN form: time of the pair GEMM + fold per
K split, per tile plan (DATR_GEMM_PLAN is read per call).   python tools/probes/gemm_tn_small.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from datr_amd import gemm  # noqa: E402
from bench_gemm import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for P, M, N in [(4400, 256, 256), (4400, 512, 256), (4400, 384, 256), (4400, 2048, 256), (4400, 256, 2048), (3600, 256, 256), (13200, 256, 256)]:
    x = torch.randn(P, N, device=dev)
    dy = torch.randn(P, M, device=dev)
    os.environ.pop("DATR_GEMM_PLAN", None)
    line = [f"P={P} M={M} N={N}: auto {timeit(lambda: gemm.gemm_tn(dy, x, bias_grad=True), 30):.1f}"]
    for tile in ["1,2,16", "1,1,16", "1,1,32"]:
        for ks in (1, 2, 4, 8, 16, 32, 64):
            os.environ["DATR_GEMM_PLAN"] = f"{tile},{ks}"
            try:
                t = timeit(lambda: gemm.gemm_tn(dy, x, bias_grad=True), 30)
            except Exception as e:  # a plan the shape does not admit
                continue
            line.append(f"{tile}/{ks} {t:.1f}")
    lib = timeit(lambda: (dy.t().mm(x), dy.sum(0)), 30)
    print("  ".join(line) + f"  | library mm + sum {lib:.1f}", flush=True)
