"""Groups the ATen element-wise / copy / fill kernels of the last steps of a bench.py kernel trace
(rocprofv3 --kernel-trace --output-format csv) by kernel and grid size: where the small-kernel time
of a step sits.   python tools/probes/elementwise_audit.py <kernel_trace.csv> [steps]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"] or "adamw_update" in r["Kernel_Name"]]
per = len(adam) // max(1, len(set(int(rows[i]["Start_Timestamp"]) // 50_000_000 for i in adam)))
# one optimizer step = a burst of FusedAdam launches; take the kernels between the bursts `steps` apart
bursts = [adam[0]]
for a, b in zip(adam, adam[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 20_000_000:
        bursts.append(b)
lo, hi = bursts[-steps - 1], bursts[-1]
agg = defaultdict(lambda: [0, 0.0])
total = 0.0
for r in rows[lo:hi]:
    k = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total += d
    if not ("at::native" in k or "rocclr" in k or "SubTensorOp" in k):
        continue
    short = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", k)[:70]
    key = (short, r.get("Grid_Size", r.get("Grid_Size_X", "")))
    agg[key][0] += 1
    agg[key][1] += d
print(f"{steps} steps, {total / steps / 1e3:.2f} ms of kernels per step")
tot = sum(v[1] for v in agg.values())
print(f"ATen / copy / MIOpen tensor-op kernels: {tot / steps / 1e3:.2f} ms per step")
for (k, g), (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{d / steps:8.1f} us/step  {n / steps:6.1f} calls/step  avg {d / n:7.1f} us  grid {g:>10}  {k}")
