"""F.linear on a 3-d input without autograd (hipBLASLt's bias-epilogue entry, with the shipped selections) against the
own NT form, per shape of the eval / two-stage passes: us, median of 20."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import gemm, tuning
tuning.enable()
dev = torch.device("cuda:0")


def med(fn, n=20):
    for _ in range(4):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[n // 2]


with torch.no_grad():
    for imgs in (4, 2):
        for K, N in ((256, 256), (256, 384), (256, 2048), (2048, 256)):
            x = torch.randn(imgs, 22223, K, device=dev)
            w, b = torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
            lib = med(lambda: torch.nn.functional.linear(x, w, b))
            own = med(lambda: gemm.gemm_nt(x.view(-1, K), w, shift=b))
            lib2 = med(lambda: torch.addmm(b, x.view(-1, K), w.t()))
            print(f"rows {imgs * 22223:6d} K {K:4d} N {N:4d}: F.linear 3-d {lib:7.1f} us | addmm 2-d {lib2:7.1f} us | own NT {own:7.1f} us")
