"""Where the slow step behind an OffsetMonitor re-measurement comes from: wall time of every step (with a
device synchronisation), of every poll / observe call, and whether the envelope changed."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from datr_amd.training import Stepper, synthetic_batch
import datr_amd.msda as M
dev = torch.device("cuda:0")
tr = Stepper(dev)
b = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
orig_poll, orig_obs = M.OffsetMonitor.poll, M.OffsetMonitor.observe


def poll(self):
    before = None if self.envelope is None else self.envelope.copy()
    t = time.perf_counter(); r = orig_poll(self); dt = time.perf_counter() - t
    changed = (before is None) != (self.envelope is None) or (before is not None and bool((before != self.envelope).any()))
    if dt > 1e-3 or changed:
        print("  poll at call", self.calls + 1, "ms", round(dt * 1e3, 2), "envelope changed", changed)
    return r


def obs(self, *a):
    t = time.perf_counter(); r = orig_obs(self, *a); dt = time.perf_counter() - t
    if dt > 1e-3:
        print("  observe at call", self.calls, "ms", round(dt * 1e3, 2))
    return r


M.OffsetMonitor.poll, M.OffsetMonitor.observe = poll, obs
for i in range(56):
    t = time.perf_counter(); tr.step(*b); torch.cuda.synchronize(); dt = time.perf_counter() - t
    if i > 3 and dt > 0.085:
        print("step", i + 1, "ms", round(dt * 1e3, 1))
