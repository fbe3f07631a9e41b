import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import fused
dev = torch.device("cuda:0")
rows = 88892
x, res, dy = (torch.randn(rows, 256, device=dev) for _ in range(3))
ln = torch.nn.LayerNorm(256).to(dev)
xg = x.clone().requires_grad_(True)
out = fused.add_layer_norm(xg, res, ln)
def med(fn, n=30):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[n // 2]
print("add_ln bwd (incl. finish) %.1f us, fwd %.1f us" % (med(lambda: torch.autograd.grad(out, xg, dy, retain_graph=True)),
                                                          med(lambda: fused.add_layer_norm(x, res, ln))))
