"""Own fp32-MFMA GEMM family (csrc/gemm_f32.hip) against the library GEMMs (hipBLASLt / rocBLAS with the
committed TunableOp selections) at the shapes of the training step: correctness against float64 on
sampled rows, then time per call.
    python tools/bench_gemm.py [--quick] [--plan tm,tn,ksplit] [--only substr]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datr_amd import gemm, tuning  # noqa: E402


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


def sample_check(got, ref_fn, M, rows=64):
    idx = torch.randint(0, M, (rows,), device=got.device)
    ref = ref_fn(idx)
    err = (got[idx].double() - ref).abs().max().item()
    return err / max(ref.abs().max().item(), 1e-30)


# (name, form, M, N, K)   M = rows (pixels / tokens), N = output features, K = reduction
P1, P2, P3, P4, TOK = 4 * 200 * 334, 4 * 100 * 167, 4 * 50 * 84, 4 * 25 * 42, 88892
FWD = [
    ("l1.conv1 256>64", P1, 64, 256), ("l1.conv3 64>256", P1, 256, 64),
    ("l2.0.conv1 256>128", P1, 128, 256), ("l2.conv1 512>128", P2, 128, 512), ("l2.conv3 128>512", P2, 512, 128),
    ("l3.0.conv1 512>256", P2, 256, 512), ("l3.conv1 1024>256", P3, 256, 1024), ("l3.conv3 256>1024", P3, 1024, 256),
    ("l4.0.conv1 1024>512", P3, 512, 1024), ("l4.conv1 2048>512", P4, 512, 2048), ("l4.conv3 512>2048", P4, 2048, 512),
    ("proj0 512>256", P2, 256, 512), ("proj1 1024>256", P3, 256, 1024), ("proj2 2048>256", P4, 256, 2048),
    ("ffn1 256>2048", TOK, 2048, 256), ("ffn2 2048>256", TOK, 256, 2048), ("lin 256>256", TOK, 256, 256),
    ("lin 256>384", TOK, 384, 256), ("dec_ffn1 256>2048", 4400, 2048, 256), ("dec_lin 256>256", 4400, 256, 256),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--no-lib", action="store_true")
    a = ap.parse_args()
    tuning.enable()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [s for s in FWD if a.only in s[0]]
    if a.quick:
        shapes = shapes[:4]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        dy = torch.randn(M, N, device=dev)
        res = torch.randn(M, N, device=dev)
        scale = torch.rand(N, device=dev) + 0.5
        shift = torch.randn(N, device=dev)
        flops = 2.0 * M * N * K
        line = {"layer": name, "M": M, "N": N, "K": K}
        # forward NT, full epilogue
        y = gemm.gemm_nt(x, w, scale=scale, shift=shift, residual=res, relu=True)
        line["err_nt"] = sample_check(y, lambda i: torch.relu(x[i].double() @ w.double().t() * scale.double() + shift.double() + res[i].double()), M)
        t = timeit(lambda: gemm.gemm_nt(x, w, scale=scale, shift=shift, residual=res, relu=True), a.iters)
        line["nt_epi_us"], line["nt_epi_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        t = timeit(lambda: gemm.gemm_nt(x, w), a.iters)
        line["nt_us"], line["nt_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        if not a.no_lib:
            t = timeit(lambda: torch.addmm(shift, x, w.t()), a.iters)
            line["lib_nt_us"], line["lib_nt_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        # data gradient NN with gate + residual + column sums
        dx, cs = gemm.gemm_nn(dy, w, residual=x, gate=x, colsum=True)
        line["err_nn"] = sample_check(dx, lambda i: (dy[i].double() @ w.double() + x[i].double()) * (x[i] > 0), M)
        ref_cs = dx.double().sum(0)
        line["err_colsum"] = ((cs.double() - ref_cs).abs().max() / ref_cs.abs().max()).item()
        t = timeit(lambda: gemm.gemm_nn(dy, w, residual=x, gate=x, colsum=True), a.iters)
        line["nn_epi_us"], line["nn_epi_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        t = timeit(lambda: gemm.gemm_nn(dy, w), a.iters)
        line["nn_us"], line["nn_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        if not a.no_lib:
            t = timeit(lambda: dy.mm(w), a.iters)
            line["lib_nn_us"], line["lib_nn_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        # weight gradient TN
        dw = gemm.gemm_tn(dy, x, rowscale=scale)
        ref = (dy.double().t() @ x.double()) * scale.double()[:, None]
        line["err_tn"] = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
        dw2 = gemm.gemm_tn(dy, x, rowscale=scale)
        line["tn_bitwise"] = bool(torch.equal(dw, dw2))
        t = timeit(lambda: gemm.gemm_tn(dy, x), a.iters)
        line["tn_us"], line["tn_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        if not a.no_lib:
            t = timeit(lambda: dy.t().mm(x), a.iters)
            line["lib_tn_us"], line["lib_tn_tf"] = round(t, 1), round(flops / t * 1e-6, 1)
        print(json.dumps(line), flush=True)
        del x, w, dy, res, y, dx, dw, dw2


if __name__ == "__main__":
    main()
