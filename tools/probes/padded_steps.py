"""Training steps on a PADDED batch only (images of (H - 32) x W inside the H x W batch tensor, mask True on the
padding: the model's general path), for a kernel trace to set beside the unpadded step's.
Usage: rocprofv3 --kernel-trace ... -- python tools/probes/padded_steps.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd import tuning  # noqa: E402
from datr_amd.training import build_training, run_steps, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
tuning.enable()
state = build_training(device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pb = synthetic_batch(2, 800 - 32, 1333, 10, dev, seed=99, channels_last=True, pad_to=(800, 1333))
run_steps(state, [pb] * 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
run_steps(state, [pb] * steps)
torch.cuda.synchronize()
print(f"padded batch: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per step")
