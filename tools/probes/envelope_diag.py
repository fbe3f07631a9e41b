"""Prints, after a few training steps at the bench shape, what every encoder layer's OffsetMonitor
measured (envelope widths per head / level) and what the pyramid plan makes of it."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import msda
from datr_amd.training import build_training, run_steps, synthetic_batch

dev = torch.device("cuda:0")
state = build_training(device=dev)
pool = [synthetic_batch(2, 800, 1333, 10, dev, seed=1 + 1000 * i) for i in range(2)]
run_steps(state, [pool[i % 2] for i in range(int(os.environ.get("STEPS", "5")))])
torch.cuda.synchronize()
shapes = torch.tensor([(100, 167), (50, 84), (25, 42), (13, 21)])
lsi = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
for i, layer in enumerate(state.model.transformer.encoder.layers):
    mon = msda._MONITORS.get(layer.self_attn)
    if mon is None:
        print("layer", i, "no monitor")
        continue
    env = mon.envelope
    print(f"layer {i}: calls {mon.calls} route {mon.route} fraction {mon.fraction:.4f} envelope {'none' if env is None else ''}")
    if env is not None:
        wy, wx = env[:, :, 1] - env[:, :, 0], env[:, :, 3] - env[:, :, 2]
        print("   width y per head (level 0):", np.round(wy[:, 0], 2).tolist())
        print("   width x per head (level 0):", np.round(wx[:, 0], 2).tolist())
        print("   max width over heads/levels:", float(max(wy.max(), wx.max())))
        print("   plan:", msda.pyramid_plan(shapes, lsi, 4, 8, 32, 4, env))
