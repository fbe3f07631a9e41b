"""GPU idle time inside a training step from a rocprofv3 kernel trace: the union of all kernel intervals of
the steady-state steps against the wall span, and where the gaps sit (which kernels surround them).
    python tools/probes/idle_gaps.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"]]
bursts = [adam[0]]
for a, b in zip(adam, adam[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 20_000_000:
        bursts.append(b)
lo, hi = bursts[-steps - 1], bursts[-1]
win = rows[lo:hi]
t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["Start_Timestamp"])
end = t0
idle = 0
gaps = []
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        idle += s - end
        gaps.append((s - end, r["Kernel_Name"][:60]))
    end = max(end, e)
print(f"{steps} steps: wall {(t1 - t0) / steps / 1e6:.2f} ms/step, GPU idle {idle / steps / 1e6:.2f} ms/step "
      f"({len(gaps) / steps:.0f} gaps/step)")
hist = defaultdict(lambda: [0, 0])
for g, n in gaps:
    b = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
    hist[b][0] += 1
    hist[b][1] += g
for b in ("<2us", "<5us", "<10us", "<50us", ">=50us"):
    print(f"  gaps {b:6s}: {hist[b][0] / steps:7.0f} per step, {hist[b][1] / steps / 1e6:6.2f} ms per step")
by = defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[n][0] += 1
    by[n][1] += g
print("idle time in front of (top 15):")
for n, (c, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {g / steps / 1e3:8.1f} us/step {c / steps:6.1f} x  {n}")
