"""Weight-gradient (TN) form of the own GEMM family: tile x split sweep per layer shape.
    python tools/probes/gemm_tn_sweep.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from datr_amd import gemm  # noqa: E402
from bench_gemm import FWD, timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, M, N, K in [s for s in FWD if only in s[0] and "ffn" not in s[0] and "l1." not in s[0]]:
    x = torch.randn(M, K, device=dev)
    dy = torch.randn(M, N, device=dev)
    line = {"layer": name, "P": M, "Cout": N, "Cin": K}
    best = (1e9, "")
    for tile in ["2,2,16", "2,2,32", "2,1,16", "1,2,16", "1,1,16"]:
        tm, tn, _ = [int(v) for v in tile.split(",")]
        tiles = ((N + 64 * tm - 1) // (64 * tm)) * ((K + 64 * tn - 1) // (64 * tn))
        for wgs in (256, 384, 512, 768, 1024, 1536):
            ks = max(1, wgs // tiles)
            os.environ["DATR_GEMM_PLAN"] = f"{tile},{ks}"
            t = timeit(lambda: gemm.gemm_tn(dy, x), 8)
            line[f"{tile}/{ks}"] = round(t, 1)
            best = min(best, (t, f"{tile}/{ks}"))
    line["best"], line["best_us"] = best[1], round(best[0], 1)
    line["best_tf"] = round(2.0 * M * N * K / best[0] * 1e-6, 1)
    print(json.dumps(line), flush=True)
