import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd.optim import FusedClipAdamW
dev = torch.device("cuda:0")
torch.manual_seed(0)
p0 = torch.randn(256, 256)
a = torch.nn.Parameter(p0.clone().to(dev)); b = torch.nn.Parameter(p0.clone().to(dev)); c = torch.nn.Parameter(p0.clone())
oa = FusedClipAdamW([a], lr=1e-4, weight_decay=1e-4)
ob = torch.optim.AdamW([b], lr=1e-4, weight_decay=1e-4, foreach=False, fused=False)
oc = torch.optim.AdamW([c], lr=1e-4, weight_decay=1e-4, foreach=False)
for s in range(4):
    g = torch.randn(256, 256, generator=torch.Generator().manual_seed(100 + s)) * 0.3
    a.grad = g.clone().to(dev); b.grad = g.clone().to(dev); c.grad = g.clone()
    oa.step(); ob.step(); oc.step()
    da = (a.detach().cpu() - c.detach()).abs().max().item()
    db = (b.detach().cpu() - c.detach()).abs().max().item()
    dm = (oa.state[a]["exp_avg"].cpu() - oc.state[c]["exp_avg"]).abs().max().item()
    dv = ((oa.state[a]["exp_avg_sq"].cpu() - oc.state[c]["exp_avg_sq"]).abs() / oc.state[c]["exp_avg_sq"]).max().item()
    print(f"step {s}: own-vs-cpu {da:.3e}  torchgpu-vs-cpu {db:.3e}  m diff {dm:.3e}  v rel diff {dv:.3e}  step={float(oa.state[a]['step'])}")
