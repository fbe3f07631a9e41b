"""Scan the device assembly of every kernel in datr_amd/csrc for memory operations the compiler serialised:
runs of (store, s_waitcnt vmcnt(<=1)) or (load, s_waitcnt vmcnt(<=1)) -- each element waiting for everything in flight,
typically bounds branches around loads / stores with other loads still pending (profiles/HISTORY.md, round 5).
Cross-compiles without a GPU.  usage: python tools/isa_scan.py [--min-run 3] [--skeleton KERNEL_SUBSTRING]"""
import argparse
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--min-run", type=int, default=3)
ap.add_argument("--skeleton", default=None, help="print the load / store / wait / MFMA skeleton of kernels whose name contains this")
args = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="isa_scan_")
for src in sorted(glob.glob(os.path.join(ROOT, "datr_amd", "csrc", "*.hip"))):
    out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(src))
    lines = open(out).read().split("\n")
    for st in [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:\s*(;.*)?$", l)]:
        en = next((i for i in range(st, len(lines)) if lines[i].strip().startswith("s_endpgm")), None)
        if en is None:
            continue
        ev = []
        for l in lines[st:en]:
            t = l.strip()
            if re.match(r"(global|buffer|flat)_(store|atomic)", t):
                ev.append("S")
            elif re.match(r"(global|buffer|flat)_load", t):
                ev.append("L")
            elif t.startswith("ds_read"):
                ev.append("r")
            elif t.startswith("ds_write"):
                ev.append("x")
            elif t.startswith("v_mfma"):
                ev.append("M")
            elif "s_waitcnt" in t and "vmcnt" in t:
                ev.append("w" if int(re.search(r"vmcnt\((\d+)\)", t).group(1)) <= 1 else "W")
            elif "s_barrier" in t:
                ev.append("|")
            elif t.startswith("s_cbranch"):
                ev.append("^")
            elif re.match(r"\.LBB", t):
                ev.append(":")
        seq = "".join(ev)
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", lines[st])[:60]
        if args.skeleton and args.skeleton in lines[st]:
            print(os.path.basename(src), name)
            print(seq)
        mem = re.sub(r"[rxM|]", "", seq)
        loads = [m.group(0).count("L") for m in re.finditer(r"(?:[\^:]*L[\^:]*w[\^:]*){%d,}" % args.min_run, mem)]
        stores = [m.group(0).count("S") for m in re.finditer(r"(?:[\^:]*S[\^:]*w[\^:]*){%d,}" % args.min_run, mem)]
        if loads or stores:
            print(f"{os.path.basename(src):20s} {name:60s} serial loads {loads} serial stores {stores}")
