"""Round 6: the own GEMM family against the library on the 88 892-row shapes, INTERLEAVED (lib, own, lib, own, ...):
back-to-back loops of one contender run into the chip's power budget (every contender converges to ~104 TF/s,
tools/probes/r06_gemm_stages.sh), so A-then-B comparisons measure the order, not the kernels.
    python tools/probes/r06_gemm_interleaved.py [--tuned]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd import gemm, tuning  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tuned", action="store_true", help="hipBLASLt with the per-shape selections of datr_amd/tuning")
ap.add_argument("--rounds", type=int, default=15)
a = ap.parse_args()
if a.tuned:
    tuning.enable()
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = 88892
for name, N, K in (("ffn1 256>2048", 2048, 256), ("ffn2 2048>256", 256, 2048), ("lin 256>256", 256, 256), ("lin 256>384", 384, 256)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * K ** -0.5
    dy = torch.randn(M, N, device=dev)
    b = torch.randn(N, device=dev)
    cands = {
        "fwd  lib addmm": lambda: torch.addmm(b, x, w.t()),
        "fwd  own nt+shift": lambda: gemm.gemm_nt(x, w, shift=b),
        "dgrad lib mm": lambda: dy.mm(w),
        "dgrad own nn": lambda: gemm.gemm_nn(dy, w),
        "wgrad lib mm": lambda: dy.t().mm(x),
        "wgrad own tn": lambda: gemm.gemm_tn(dy, x),
    }
    times = {k: [] for k in cands}
    for k, f in cands.items():
        f()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for k, f in cands.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); f(); e.record()
            times[k].append((s, e))
    torch.cuda.synchronize()
    for k, ev in times.items():
        us = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
        med = us[len(us) // 2]
        print(f"{name:16s} {k:18s} median {med:8.1f} us  {2.0 * M * N * K / med * 1e-6:6.1f} TF/s   min {us[0]:8.1f}")
