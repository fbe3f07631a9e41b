"""cProfile of the host side of one training step (what the CPU spends its ~98 ms on)."""
import os, sys, cProfile, pstats, argparse, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import Stepper, synthetic_batch
dev = torch.device("cuda:0")
tr = Stepper(dev)
samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(6):
    tr.step(samples, targets)
torch.cuda.synchronize()
import time
# phase split: forward / criterion / backward / optimizer (host enqueue time, GPU not waited for)
from datr_amd.criterion import weighted_total
T = [0.0] * 5
for _ in range(10):
    t0 = time.perf_counter(); out = tr.model(samples, targets)
    t1 = time.perf_counter(); ld = tr.criterion(out, targets); loss = weighted_total(ld, tr.criterion.weight_dict)
    t2 = time.perf_counter(); tr.optimizer.zero_grad(); loss.backward()
    t3 = time.perf_counter(); torch.nn.utils.clip_grad_norm_(tr.model.parameters(), tr.state.cfg.clip_max_norm)
    t4 = time.perf_counter(); tr.optimizer.step()
    t5 = time.perf_counter()
    for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
        T[i] += (b - a) * 100
    torch.cuda.synchronize()
print("host ms/step: forward %.1f  criterion %.1f  backward %.1f  clip %.1f  optimizer %.1f" % tuple(T))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    out = tr.model(samples, targets)
    ld = tr.criterion(out, targets)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().split("\n")[:60]))
