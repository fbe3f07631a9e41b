"""Steady-state per-kernel summary from a rocprofv3 kernel trace.

rocprofv3's own --stats table covers the whole process, including MIOpen's find phase in the
first step.  This script keeps only the kernels of the last `--steps` training steps of a
bench.py run (steps are delimited by the AdamW kernel, one launch group per step) and writes a
compact CSV:  name, calls/step, total ms/step, avg us, percent.

    python tools/kstats.py /tmp/prof/step_kernel_trace.csv --steps 2 --out profiles/x.csv
"""
import argparse
import csv
import re
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:150]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--marker", default="msda_fwd_rows")
    ap.add_argument("--per-step", type=int, default=24, help="marker launches per step")
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--split", default=None,
                    help="also print this kernel's launches grouped by grid size (e.g. msda_fwd_rows: "
                         "encoder and decoder calls share a kernel name)")
    ap.add_argument("--split-out", default=None)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    ks, ke, kn = "Start_Timestamp", "End_Timestamp", "Kernel_Name"
    rows.sort(key=lambda r: int(r[ks]))
    marks = [i for i, r in enumerate(rows) if a.marker in r[kn]]
    n_steps = len(marks) // a.per_step
    assert n_steps > a.steps, f"only {n_steps} steps in trace"
    first = marks[(n_steps - a.steps) * a.per_step]
    # a step starts a little before its first MSDA launch (backbone comes first): take the
    # window between the first marker of step (n-steps-1)+1 ... use marker-to-marker periods
    start_idx = marks[(n_steps - a.steps - 1) * a.per_step]
    end_idx = marks[(n_steps - 1) * a.per_step]
    window = rows[start_idx:end_idx]          # exactly `steps` periods of the step cycle
    t0, t1 = int(window[0][ks]), int(window[-1][ks])
    # GPU busy time = union of kernel intervals in the window
    busy, cur_s, cur_e = 0, None, None
    for r in window:
        a_, b_ = int(r[ks]), int(r[ke])
        if cur_e is None or a_ > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = a_, b_
        else:
            cur_e = max(cur_e, b_)
    if cur_e is not None:
        busy += cur_e - cur_s
    agg = defaultdict(lambda: [0, 0])
    for r in window:
        d = int(r[ke]) - int(r[ks])
        e = agg[short(r[kn])]
        e[0] += 1
        e[1] += d
    total = sum(v[1] for v in agg.values())
    wall = (t1 - t0) / a.steps / 1e6
    out = [("name", "calls_per_step", "ms_per_step", "avg_us", "percent_of_kernel_time")]
    for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((name, f"{c / a.steps:.1f}", f"{d / a.steps / 1e6:.3f}", f"{d / c / 1e3:.1f}",
                    f"{100 * d / total:.2f}"))
    print(f"# gpu_busy_ms_per_step={busy / a.steps / 1e6:.1f} (idle {wall - busy / a.steps / 1e6:.1f})")
    print(f"# steps={a.steps} wall_ms_per_step={wall:.1f} kernel_ms_per_step={total / a.steps / 1e6:.1f} "
          f"launches_per_step={len(window) / a.steps:.0f} distinct={len(agg)}")
    for row in out[:a.top + 1]:
        print(", ".join(row))
    if a.split:
        groups = defaultdict(list)
        gk = next((k for k in ("Grid_Size", "Grid_Size_X", "grid_size") if k in window[0]), None)
        for r in window:
            if a.split in r[kn]:
                groups[r[gk] if gk else "?"].append((int(r[ke]) - int(r[ks])) / 1e3)
        lines = [f"# {a.split}: launches in the steady-state window grouped by grid size ({gk})",
                 "grid_size,launches_per_step,avg_us,min_us,max_us"]
        for g, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{g},{len(v) / a.steps:.1f},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}")
        print("\n".join(lines))
        if a.split_out:
            open(a.split_out, "w").write("\n".join(lines) + "\n")
    if a.out:
        with open(a.out, "w") as f:
            f.write(f"# gpu_busy_ms_per_step={busy / a.steps / 1e6:.1f}\n")
            f.write(f"# steady-state steps={a.steps} wall_ms_per_step={wall:.1f} "
                    f"kernel_ms_per_step={total / a.steps / 1e6:.1f} "
                    f"launches_per_step={len(window) / a.steps:.0f}\n")
            w = csv.writer(f)
            w.writerows(out)


if __name__ == "__main__":
    main()
