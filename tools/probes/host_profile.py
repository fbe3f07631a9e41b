"""cProfile of the host side of a training step (the GPU runs asynchronously): where the ~78 ms of enqueue
time per step go.   python tools/probes/host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import Stepper, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tr = Stepper(dev)
samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(6):
    tr.step(samples, targets)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.step(samples, targets)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
