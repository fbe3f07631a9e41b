"""A/B of the encoder call's backward: datr_msda_backward_pyramid_query_f32 (one pass, the query projection's
gradient rows written by the LDS-window kernel) against datr_msda_backward_pyramid_f32 + the prologue's backward
kernel, at the training step's size (N = 4, 1333 x 800), for the ring offsets (two tasks per wave) and for
N(0, 2.5 px) offsets (three tasks per wave).  Usage: python tools/probes/query_grad_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import msda as M  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(100, 167), (50, 84), (25, 42), (13, 21)]
N, Mh, D, P = 4, 8, 32, 4
S = sum(h * w for h, w in shapes)
sh = torch.tensor(shapes, device=dev)
lsi = torch.cat([sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]])
g = torch.Generator(device=dev).manual_seed(1)
value = torch.randn(N, S, Mh, D, device=dev, generator=g)
attn = torch.softmax(torch.randn(N, S, Mh, 16, device=dev, generator=g), -1).view(N, S, Mh, 4, P)
go = torch.randn(N, S, Mh * D, device=dev, generator=g)
centres = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"), -1)
                     .flip(-1).reshape(-1, 2) for h, w in shapes], 0)
wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=dev).view(1, 1, 1, 4, 1, 2)
ring = M.MSDeformAttn(256, 4, Mh, P).sampling_offsets.bias.detach().view(1, 1, Mh, 4, P, 2).to(dev)
ref2 = torch.zeros(N, S, 4, 2, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for kind in ("ring", "gauss2.5"):
    if kind == "ring":
        loc = (centres.view(1, S, 1, 1, 1, 2) + ring / wh).expand(N, S, Mh, 4, P, 2).contiguous()
    else:
        loc = (centres.view(1, S, 1, 1, 1, 2) + 2.5 * torch.randn(N, S, Mh, 4, P, 2, device=dev, generator=g) / wh).contiguous()
    env = M.measure_envelope(loc, sh)
    plan = M.pyramid_plan(sh, lsi, N, Mh, D, P, env)

    def two_pass():
        gv, gl, ga = M.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64, envelope=env)
        return M._prologue_backward(gl, ga, attn, ref2, (N, S, 384))

    def one_pass():
        return M.ms_deform_attn_backward_query_grad(value, sh, lsi, loc, attn, go, 64, envelope=env)

    assert one_pass() is not None
    print(f"{kind}: tasks per wave {plan['tasks_per_wave']}, grid {plan['grid']}: two passes {timed(two_pass):.1f} us, "
          f"query-gradient rows {timed(one_pass):.1f} us")
