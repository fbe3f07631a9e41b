"""Plan sweep of the own GEMM family (csrc/gemm_f32.hip): every (tile, step) plan per layer shape and
operand form, library time beside it.  Output: one JSON line per (layer, form).
    python tools/sweep_gemm.py [--only substr]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import gemm, tuning  # noqa: E402
from bench_gemm import FWD, timeit  # noqa: E402

PLANS = ["2,2,32", "2,2,16", "2,1,32", "2,1,16", "1,2,16", "1,1,16", "1,1,32"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--epi", action="store_true")
    a = ap.parse_args()
    tuning.enable()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, M, N, K in [s for s in FWD if a.only in s[0]]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        dy = torch.randn(M, N, device=dev)
        res = torch.randn(M, N, device=dev)
        sc = torch.rand(N, device=dev)
        flops = 2.0 * M * N * K
        forms = {
            "nt": (lambda: gemm.gemm_nt(x, w), lambda: x.mm(w.t())),
            "nt_bias": (lambda: gemm.gemm_nt(x, w, shift=sc), lambda: torch.addmm(sc, x, w.t())),
            "nn": (lambda: gemm.gemm_nn(dy, w), lambda: dy.mm(w)),
            "tn": (lambda: gemm.gemm_tn(dy, x), lambda: dy.t().mm(x)),
        }
        if a.epi:
            forms = {
                "nt_res_relu": (lambda: gemm.gemm_nt(x, w, scale=sc, shift=sc, residual=res, relu=True), lambda: x.mm(w.t())),
                "nn_res_gate": (lambda: gemm.gemm_nn(dy, w, residual=x, gate=x), lambda: dy.mm(w)),
                "nn_gate_cs": (lambda: gemm.gemm_nn(dy, w, gate=x, colsum=True), lambda: dy.mm(w)),
            }
        for form, (own, lib) in forms.items():
            line = {"layer": name, "form": form, "M": M, "N": N, "K": K, "lib_us": round(timeit(lib, a.iters), 1)}
            for plan in PLANS:
                tm, tn, bk = plan.split(",")
                if (tn == "2" and N <= 64):
                    continue
                os.environ["DATR_GEMM_PLAN"] = plan + ",0"
                line[plan] = round(timeit(own, a.iters), 1)
            os.environ.pop("DATR_GEMM_PLAN", None)
            best = min((v, k) for k, v in line.items() if "," in k)
            line["best"], line["best_us"], line["best_tf"] = best[1], best[0], round(flops / best[0] * 1e-6, 1)
            line["lib_tf"] = round(flops / line["lib_us"] * 1e-6, 1)
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
