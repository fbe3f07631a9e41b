"""One (layer shape, form, plan, epilogue) of the own GEMM family, repeated -- the target of PMC passes.
    python tools/probes/gemm_one.py --layer ffn1 --form nn --plan 1,2,16 [--epi gate_cs] [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from datr_amd import gemm  # noqa: E402
from bench_gemm import FWD, timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="ffn1")
ap.add_argument("--form", default="nn")
ap.add_argument("--plan", default="")
ap.add_argument("--epi", default="none")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
if a.plan:
    os.environ["DATR_GEMM_PLAN"] = a.plan + ",0"
name, M, N, K = [s for s in FWD if a.layer in s[0]][0]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * K ** -0.5
dy = torch.randn(M, N, device=dev)
res = torch.randn(M, N, device=dev)
sc = torch.rand(N, device=dev)
fns = {
    ("nt", "none"): lambda: gemm.gemm_nt(x, w),
    ("nt", "res_relu"): lambda: gemm.gemm_nt(x, w, scale=sc, shift=sc, residual=res, relu=True),
    ("nn", "none"): lambda: gemm.gemm_nn(dy, w),
    ("nn", "res_gate"): lambda: gemm.gemm_nn(dy, w, residual=x, gate=x),
    ("nn", "gate_cs"): lambda: gemm.gemm_nn(dy, w, gate=x, colsum=True),
    ("tn", "none"): lambda: gemm.gemm_tn(dy, x),
    ("lib_nn", "none"): lambda: dy.mm(w),
    ("lib_nt", "none"): lambda: x.mm(w.t()),
}
t = timeit(fns[(a.form, a.epi)], a.iters)
print(f"{name} {a.form} {a.epi} plan={a.plan or 'auto'}: {t:.1f} us = {2.0 * M * N * K / t * 1e-6:.1f} TF/s")
