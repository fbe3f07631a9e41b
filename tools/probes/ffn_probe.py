import torch, time, sys, os
sys.path.insert(0, "/root/repo")
from datr_amd import tuning
tuning.enable()
dev = torch.device("cuda:0")
R = 88892
x = torch.randn(R, 256, device=dev)
W1 = torch.randn(2048, 256, device=dev) * 0.05
b1 = torch.randn(2048, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("linear            us", t(lambda: torch.nn.functional.linear(x, W1, b1)))
print("linear+relu       us", t(lambda: torch.relu(torch.nn.functional.linear(x, W1, b1))))
print("_addmm_activation us", t(lambda: torch._addmm_activation(b1, x, W1.t(), use_gelu=False)))
y1 = torch.relu(torch.nn.functional.linear(x, W1, b1)); y2 = torch._addmm_activation(b1, x, W1.t(), use_gelu=False)
print("maxdiff", (y1 - y2).abs().max().item())
h = y1; dh = torch.randn_like(h)
print("threshold_bwd     us", t(lambda: torch.ops.aten.threshold_backward(dh, h, 0)))
dz = torch.ops.aten.threshold_backward(dh, h, 0)
print("sum(0)            us", t(lambda: dz.sum(0)))
