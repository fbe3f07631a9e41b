"""Does the vector-memory path charge a gather per LANE or per DISTINCT cache line?
Row kernel (DATR_MSDA_PYR_FWD=0), N=4 encoder call, three location sets:
  model   -- pixel-centre reference points + the initial offset ring (all rows distinct)
  coarse  -- every level's samples computed as usual, but the queries of a wave (8 consecutive
             pixels) share ONE reference point (the first query's): 8x fewer distinct rows
  const   -- every sample at the image centre: one distinct row per level
"""
import os, sys, json
os.environ["DATR_MSDA_PYR_FWD"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_msda as B
from datr_amd import msda
dev = torch.device("cuda:0")
value, sh, lsi, loc, attn = B.make_inputs(dev, 22223, "model", N=4)
def t(l):
    f = lambda: msda.ms_deform_attn_forward(value, sh, lsi, l, attn, 64)
    return B.time_fn(f, 30)[0]
res = {"model": t(loc)}
S = loc.shape[1]
idx = (torch.arange(S, device=dev) // 8) * 8
res["shared_ref_8"] = t(loc[:, idx].contiguous())
idx = (torch.arange(S, device=dev) // 32) * 32
res["shared_ref_32"] = t(loc[:, idx].contiguous())
res["const"] = t(torch.full_like(loc, 0.5))
print(json.dumps(res))
