"""Host cost of a NEW offset envelope: the first forward / backward call with it (plan search, possibly another
kernel variant's first launch) against the calls after it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from datr_amd import msda
from bench_msda import make_inputs
dev = torch.device("cuda:0")
value, sh, lsi, loc, attn = make_inputs(dev, 22223, "model", N=4)
go = torch.randn(4, 22223, 256, device=dev)
env0 = msda.measure_envelope(loc, sh)
for k in range(4):
    env = env0.copy()
    env[..., 1::2] += 0.25 * k
    env[..., 0::2] -= 0.25 * (k // 2)
    for name, fn in (("fwd", lambda: msda.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64, envelope=env)),
                     ("bwd", lambda: msda.ms_deform_attn_backward(value, sh, lsi, loc, attn, go, 64, envelope=env))):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        print(f"envelope {k} {name}: first call {ts[0]:.2f} ms, then {ts[1]:.2f} {ts[2]:.2f} ms; plan {msda.pyramid_plan(sh, lsi, 4, 8, 32, 4, env)['grid'] if name == 'fwd' else ''}")
