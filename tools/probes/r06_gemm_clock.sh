#!/bin/bash
# Round 6: is the own GEMM family schedule-bound or power-bound?  For one shape (ffn2 forward: M = 88 892, N = 256, K = 2048,
# 93.2 GFLOP) and three contenders -- tuned library, own shipped plan, own pipelined 256 x 128 plan -- the matrix-pipe busy
# cycles, the elapsed cycles (GRBM_GUI_ACTIVE / 8 XCDs) and the kernel duration: busy fraction = MFMA busy / (elapsed x 1024
# SIMDs ... counted per SIMD), effective clock = elapsed cycles / duration.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
LAYER=${1:-ffn2}
for who in lib auto pipe; do
  case $who in
    lib)  ARGS="--form lib_nt"; unset DATR_GEMM_PLAN;;
    auto) ARGS="--form nt"; unset DATR_GEMM_PLAN;;
    pipe) ARGS="--form nt --plan 4,2,16";;
  esac
  for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
    rm -rf /tmp/gc_$who_$C
    (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/gc_${who}_$C -o run -- python /root/repo/tools/probes/gemm_one.py --layer $LAYER $ARGS --iters 8 > /tmp/gc.log 2>&1) || true
  done
  python - $who <<'PY'
import csv, glob, sys, collections
who = sys.argv[1]
vals, dur = {}, []
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
    f = glob.glob(f"/tmp/gc_{who}_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])) if f else []:
        if r["Counter_Name"] == c and ("gemm_f32_kernel" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]):
            agg[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
    k = max(agg, key=lambda k_: len(agg[k_])) if agg else None
    vals[c] = sum(agg[k]) / len(agg[k]) if k else float("nan")
    if c == "GRBM_GUI_ACTIVE" and f:
        t = glob.glob(f"/tmp/gc_{who}_{c}/**/*kernel_trace.csv", recursive=True)
        for r in csv.DictReader(open(t[0])) if t else []:
            if k and r["Kernel_Name"][:50] == k:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        name = k
d = sorted(dur)[len(dur) // 2] if dur else float("nan")
elapsed = vals["GRBM_GUI_ACTIVE"] / 8
print(f"{who:5s} {name}: duration {d:7.1f} us, elapsed {elapsed:9.0f} cycles -> clock {elapsed / d / 1e3:5.2f} GHz, "
      f"MFMA busy {vals['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} -> per SIMD {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / elapsed:6.1%} of the elapsed cycles")
PY
done
