"""Rounds audit of a rocprofv3 kernel trace: for every (kernel, grid) of the steady-state steps, the workgroups per CU the
kernel's registers / LDS / threads admit, the number of rounds its grid makes over the chip's 256 CUs, and how full the last
round is -- a kernel whose 1 120 workgroups run on 512 slots pays three rounds for 2.2 (mha_bwd before round 5).
    python tools/occupancy_audit.py <kernel_trace.csv> [--min-ms 0.2]"""
import argparse, collections, csv, math

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--min-ms", type=float, default=0.15)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
agg = collections.defaultdict(lambda: [0, 0.0, None])
for r in rows:
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    nwg = grid // max(wg, 1)
    vg = int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
    lds = int(r.get("LDS_Block_Size", 0) or 0)
    key = (r["Kernel_Name"][:70], nwg, wg, vg, lds)
    e = agg[key]
    e[0] += 1
    e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = []
for (name, nwg, wg, vg, lds), (n, us, _) in agg.items():
    waves = max(1, wg // 64)
    alloc = max(8, math.ceil(max(vg, 1) / 8) * 8)
    by_reg = min(8, 512 // alloc) * 4 // waves if waves <= 4 * min(8, 512 // alloc) else 0
    by_lds = (160 * 1024) // lds if lds else 99
    by_thr = 2048 // wg
    per_cu = max(1, min(by_reg, by_lds, by_thr, 16))
    slots = per_cu * 256
    rounds = nwg / slots
    full = math.ceil(rounds)
    waste = 1 - rounds / full
    ms = us / 1e3 / a.steps
    if ms >= a.min_ms:
        out.append((ms, name, n, nwg, wg, vg, lds, per_cu, rounds, waste))
print(f"{'ms/step':>8} {'launches':>8} {'WGs':>7} {'thr':>4} {'regs':>4} {'LDS':>6} {'WG/CU':>5} {'rounds':>7} {'last round empty':>9}  kernel")
for ms, name, n, nwg, wg, vg, lds, per_cu, rounds, waste in sorted(out, reverse=True):
    flag = " <--" if waste > 0.25 and rounds < 6 else ""
    print(f"{ms:8.3f} {n:8d} {nwg:7d} {wg:4d} {vg:4d} {lds:6d} {per_cu:5d} {rounds:7.2f} {100 * waste:8.0f} %  {name}{flag}")
