"""cProfile of the host side of the training step at a size where the host is the bound (2 pairs of 480 x 800):
where the ~40 ms of enqueue work per step go.  Usage: python tools/probes/host_profile.py [top]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import tuning  # noqa: E402
from datr_amd.training import build_training, run_steps, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tuning.enable()
state = build_training(device=dev)
b = synthetic_batch(2, 480, 800, 10, dev, seed=1)
run_steps(state, [b] * 5)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_steps(state, [b] * 10)
torch.cuda.synchronize()
print(f"480 x 800: {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms per step")
pr = cProfile.Profile()
pr.enable()
run_steps(state, [b] * 10)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
top = int(sys.argv[1]) if len(sys.argv) > 1 else 35
st.sort_stats("tottime").print_stats(top)
st.sort_stats("cumulative").print_stats(r"datr_amd|torch/autograd/function|engine", top)
