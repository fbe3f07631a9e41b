"""Round 6: the software-pipelined 256 x 128 plan of the own GEMM family (DATR_GEMM_PLAN=4,2,16,0) -- correctness against
float64 on sampled rows (every form, with epilogues), then timings interleaved per launch with the shipped plan and the tuned
library on the 88 892-row shapes.
    python tools/probes/r06_gemm_pipe.py [--rounds 12]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import gemm, tuning  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--no-check", action="store_true")
a = ap.parse_args()
tuning.enable()
dev = torch.device("cuda:0")
torch.manual_seed(0)
PIPE = "4,2,16,0"


def with_plan(plan, fn):
    def run():
        if plan:
            os.environ["DATR_GEMM_PLAN"] = plan
        else:
            os.environ.pop("DATR_GEMM_PLAN", None)
        try:
            return fn()
        finally:
            os.environ.pop("DATR_GEMM_PLAN", None)
    return run


def check(name, got, ref_rows, idx, tol=2e-5):
    err = (got[idx].double() - ref_rows).abs().max().item() / max(ref_rows.abs().max().item(), 1e-30)
    print(f"  check {name:28s} max rel err {err:.2e} {'OK' if err < tol else 'WRONG'}")
    assert err < tol, name


M = 88892
if not a.no_check:
    for (Mc, N, K) in ((88892, 2048, 256), (88892, 256, 2048), (4013, 384, 256), (66800, 512, 256)):
        x = torch.randn(Mc, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
        dy = torch.randn(Mc, N, device=dev); b = torch.randn(N, device=dev); sc = torch.rand(N, device=dev) + 0.5
        res = torch.randn(Mc, N, device=dev); gate = torch.randn(Mc, K, device=dev)
        idx = torch.randint(0, Mc, (97,), device=dev)
        idx[0], idx[1] = 0, Mc - 1
        print(f"M={Mc} N={N} K={K}")
        y = with_plan(PIPE, lambda: gemm.gemm_nt(x, w, shift=b))()
        check("nt + shift", y, x[idx].double() @ w.double().t() + b.double(), idx)
        y = with_plan(PIPE, lambda: gemm.gemm_nt(x, w, scale=sc, shift=b, residual=res, relu=True))()
        check("nt scale/shift/res/relu", y, torch.relu((x[idx].double() @ w.double().t()) * sc.double() + b.double() + res[idx].double()), idx)
        dx, cs = with_plan(PIPE, lambda: gemm.gemm_nn(dy, w, gate=gate, colsum=True))()
        full = (dy.double() @ w.double()) * (gate > 0)
        check("nn + gate", dx, full[idx], idx)
        e = (cs.double() - full.sum(0)).abs().max().item() / full.sum(0).abs().max().item()
        print(f"  check nn colsum                    max rel err {e:.2e}")
        assert e < 1e-4
        dw, bg = with_plan(PIPE, lambda: gemm.gemm_tn(dy, x, bias_grad=True))()
        ridx = torch.arange(0, N, max(1, N // 64), device=dev)
        check("tn (weight gradient)", dw, (dy[:, ridx].double().t() @ x.double()), ridx, tol=1e-4)
        e = (bg.double() - dy.double().sum(0)).abs().max().item() / dy.double().sum(0).abs().max().item()
        print(f"  check tn bias gradient             max rel err {e:.2e}")
        assert e < 1e-4
        del x, w, dy, res, gate, full

for name, N, K in (("ffn1 256>2048", 2048, 256), ("ffn2 2048>256", 256, 2048), ("lin 256>256", 256, 256), ("lin 256>384", 384, 256)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
    dy = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev); gate = torch.randn(M, K, device=dev)
    cands = {
        "fwd   lib addmm": lambda: torch.addmm(b, x, w.t()),
        "fwd   own auto": with_plan(None, lambda: gemm.gemm_nt(x, w, shift=b)),
        "fwd   own PIPE": with_plan(PIPE, lambda: gemm.gemm_nt(x, w, shift=b)),
        "dgrad lib mm": lambda: dy.mm(w),
        "dgrad own auto": with_plan(None, lambda: gemm.gemm_nn(dy, w)),
        "dgrad own PIPE": with_plan(PIPE, lambda: gemm.gemm_nn(dy, w)),
        "dgrad own auto +gate+colsum": with_plan(None, lambda: gemm.gemm_nn(dy, w, gate=gate, colsum=True)),
        "dgrad own PIPE +gate+colsum": with_plan(PIPE, lambda: gemm.gemm_nn(dy, w, gate=gate, colsum=True)),
        "wgrad lib mm": lambda: dy.t().mm(x),
        "wgrad own auto": with_plan(None, lambda: gemm.gemm_tn(dy, x)),
        "wgrad own PIPE": with_plan(PIPE, lambda: gemm.gemm_tn(dy, x)),
    }
    times = {k: [] for k in cands}
    for f in cands.values():
        f()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for k, f in cands.items():
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); f(); e_.record()
            times[k].append((s_, e_))
    torch.cuda.synchronize()
    for k, ev in times.items():
        us = sorted(s_.elapsed_time(e_) * 1e3 for s_, e_ in ev)
        med = us[len(us) // 2]
        print(f"{name:16s} {k:30s} median {med:8.1f} us  {2.0 * M * N * K / med * 1e-6:6.1f} TF/s   min {us[0]:8.1f}")
