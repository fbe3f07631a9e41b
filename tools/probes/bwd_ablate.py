import os, sys, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_msda as B
from datr_amd import msda
dev = torch.device("cuda:0")
value, sh, lsi, loc, attn = B.make_inputs(dev, 22223, "model")
go = torch.randn(2, 22223, 256, device=dev)
f = lambda: msda.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64)
m, mn = B.time_fn(f, 30)
print(os.environ.get("DATR_MSDA_ABLATE", "0"), round(m, 1), round(mn, 1))
