"""GEMMs of one training step by shape: calls, device time, achieved TF/s (torch profiler, record_shapes)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from datr_amd.training import Stepper, synthetic_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tr = Stepper(dev)
samples, targets = synthetic_batch(2, 800, 1333, 10, dev, seed=1)
for _ in range(4):
    tr.step(samples, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(samples, targets)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::_addmm_activation", "aten::bmm", "aten::addmm_"):
        sh = e.input_shapes
        try:
            if e.key == "aten::mm":
                (m, k), (_, n) = sh[0], sh[1]
            elif e.key == "aten::bmm":
                (b, m, k), (_, _, n) = sh[0], sh[1]; m *= b
            else:
                (m, k), (_, n) = sh[1], sh[2]
        except Exception:
            continue
        fl = 2.0 * m * n * k * e.count
        rows.append((e.self_device_time_total, e.count, e.key, m, n, k, fl / max(e.self_device_time_total, 1e-9) / 1e6))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"GEMM device time {tot / 1e3:.2f} ms")
for t, c, key, m, n, k, tf in rows[:40]:
    print(f"{t / 1e3:7.3f} ms {c:4d}x {key:24s} M={m:6d} N={n:5d} K={k:6d}  {tf:6.1f} TF/s  {t / c:7.1f} us each")
