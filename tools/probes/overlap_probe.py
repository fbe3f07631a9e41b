"""Do an MFMA-bound weight-gradient GEMM and the LDS/VALU-bound MSDA kernels overlap when they are issued on two
streams?  (The FFN weight gradients dW1 = dh^T x, dW2 = dsum^T h are off the backward's dependency chain: nothing
downstream of them until the optimizer.)  Sequential on one stream vs the GEMMs on a side stream, wall time of
10 rounds by HIP events on the main stream after it has waited for the side stream."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from datr_amd import msda, tuning, fused
import bench_msda
tuning.enable()
dev = torch.device("cuda:0")
rows = 88892
dh, h = torch.randn(rows, 2048, device=dev), torch.randn(rows, 2048, device=dev)
x, dy = torch.randn(rows, 256, device=dev), torch.randn(rows, 256, device=dev)
value, sh, lsi, loc, attn = bench_msda.make_inputs(dev, 22223, "model", N=4)
go = torch.randn(4, 22223, 256, device=dev)
env = msda.measure_envelope(loc, sh)
y, res, gamma = torch.randn(rows, 256, device=dev), torch.randn(rows, 256, device=dev), torch.ones(256, device=dev)
ln = torch.nn.LayerNorm(256).to(dev)


def gemms():
    return dh.t().mm(x), dy.t().mm(h)


def others():
    msda.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64, envelope=env)
    msda.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64, envelope=env)
    out = fused.add_layer_norm(y.requires_grad_(True), res, ln)
    out.backward(dy)


side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    main.wait_stream(side)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


def seq():
    gemms(); others()


def conc():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gemms()
    others()
    main.wait_stream(side)


tg, to = timed(gemms), timed(others)
ts, tc = timed(seq), timed(conc)
print(f"two weight-gradient GEMMs alone {tg:.0f} us; MSDA bwd + fwd + add_ln fwd/bwd alone {to:.0f} us; "
      f"sequential {ts:.0f} us; GEMMs on a side stream {tc:.0f} us  (saved {ts - tc:.0f} us = {100 * (ts - tc) / to:.0f} % of the non-MFMA work)")
