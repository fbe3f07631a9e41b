"""What the stride-2 3x3 convolutions would cost as gathered GEMMs of the own family: plain GEMMs of the same
M / N / K (K = 9 Cin forward, the parity classes' tap counts for the data gradient, P for the weight gradient)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from datr_amd import gemm
from bench_gemm import timeit
dev = torch.device("cuda:0")
for name, P, C, Co in [("layer2.0.conv2", 4 * 100 * 167, 128, 128), ("layer3.0.conv2", 4 * 50 * 84, 256, 256),
                       ("layer4.0.conv2", 4 * 25 * 42, 512, 512), ("input_proj3", 4 * 13 * 21, 2048, 256)]:
    x = torch.randn(P, 9 * C, device=dev); w = torch.randn(9 * C, Co, device=dev)
    t_f = timeit(lambda: gemm.gemm_nn(x, w), 10)
    # data gradient: 4 P input pixels, 9/4 taps each on average: K = 2.25 Co
    dy = torch.randn(4 * P, (9 * Co) // 4 // 32 * 32, device=dev); wt = torch.randn(dy.shape[1], C, device=dev)
    t_d = timeit(lambda: gemm.gemm_nn(dy, wt), 10)
    # weight gradient: 9 products [Co, P] x [P, C]
    g = torch.randn(P, Co, device=dev); xx = torch.randn(P, 9 * C, device=dev)
    t_w = timeit(lambda: gemm.gemm_tn(g, xx), 10)
    print(f"{name}: forward {t_f:.0f} us  dgrad {t_d:.0f} us  wgrad {t_w:.0f} us")
