import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.fused import add_layer_norm
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
norm = torch.nn.LayerNorm(256).to(dev)
for rows in (88892, 4400):
    x = torch.randn(rows, 256, device=dev, requires_grad=True)
    r = torch.randn(rows, 256, device=dev, requires_grad=True)
    go = torch.randn(rows, 256, device=dev)
    print(rows, "fwd fused", round(t(lambda: add_layer_norm(x, r, norm)), 1), "torch", round(t(lambda: norm(x + r)), 1))
    y1 = add_layer_norm(x, r, norm); y2 = norm(x + r)
    print(rows, "bwd fused", round(t(lambda: torch.autograd.grad(y1, (x, r, norm.weight, norm.bias), go, retain_graph=True)), 1),
          "torch", round(t(lambda: torch.autograd.grad(y2, (x, r, norm.weight, norm.bias), go, retain_graph=True)), 1))
