"""Ordered kernel sequence between two occurrences of a marker kernel in a rocprofv3 kernel trace (one encoder
layer's forward or backward, say): name, duration, gap to the previous kernel's end.
usage: python tools/kseq.py <kernel_trace.csv> --marker msda_bwd_pyr_d32 --occurrence 40 [--count 1]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", required=True)
ap.add_argument("--occurrence", type=int, default=30, help="start at this occurrence of the marker (0-based)")
ap.add_argument("--count", type=int, default=1, help="how many marker-to-marker windows")
args = ap.parse_args()
rows = []
with open(args.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
marks = [i for i, r in enumerate(rows) if args.marker in r[2]]
a, b = marks[args.occurrence], marks[args.occurrence + args.count]
prev_end = rows[a - 1][1] if a else rows[a][0]
total = 0
for s, e, name, grid in rows[a:b + 1]:
    print(f"{(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  grid {grid:>9s}  {name[:110]}")
    prev_end = max(prev_end, e)
    total += e - s
print(f"window: {(rows[b][0] - rows[a][0]) / 1e3:.1f} us wall, {total / 1e3:.1f} us of kernels")
