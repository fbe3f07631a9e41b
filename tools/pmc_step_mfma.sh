#!/bin/bash
# In-step matrix-pipe utilisation by kernel family: SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE of
# every kernel of a few bench.py training steps (rocprofv3 PMC pass with --kernel-trace only).
#   usage: tools/pmc_step_mfma.sh <out.txt>      (run on the GPU box from the repo root)
set -e
OUT=${1:-gpurun_out/step_mfma.txt}
REPO=$PWD
export TMPDIR=/tmp
mkdir -p "$(dirname "$OUT")"
rm -rf /tmp/pmc_step
cd /tmp
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d /tmp/pmc_step -o run -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 \
    > /tmp/pmc_step.log 2>&1 || { tail -3 /tmp/pmc_step.log; exit 1; }
cd $REPO
python - "$(find /tmp/pmc_step -name '*counter_collection.csv' | head -1)" > "$OUT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(dict)                      # dispatch id -> counter -> value
name = {}
for r in rows:
    k = r["Dispatch_Id"]
    per[k][r["Counter_Name"]] = float(r["Counter_Value"])
    name[k] = r["Kernel_Name"]
def family(n):
    if n.startswith("Cijk_"): return "hipBLASLt / rocBLAS GEMMs"
    if "wino_wgrad_nhwc" in n: return "own Winograd weight gradient (wino_wgrad.hip)"
    if "wino_conv" in n: return "own Winograd convolutions (wino.hip)"
    if "gemm_f32_kernel" in n: return "own GEMM family (gemm_f32.hip: 1x1 convs, FFN dz, weight gradients)"
    if "tap_wgrad(" in n: return "own stride-2 weight gradient (conv_tap.hip)"
    if "tap_conv" in n: return "own stride-2 convolutions fwd + dgrad (conv_tap.hip)"
    if "stem_conv" in n: return "own stem convolution (stem.hip)"
    if "mha_" in n: return "own attention fwd + bwd (mha_fwd.hip, mha_bwd.hip)"
    if "igemm_wrw" in n or "bwd_weight" in n: return "MIOpen convolution weight gradients"
    if "igemm_fwd" in n or "conv_fwd" in n or "Sp3AsmConv" in n: return "MIOpen convolution forward"
    if "igemm_bwd" in n: return "MIOpen convolution data gradients"
    return None
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
tot = [0.0, 0.0]
for k, c in per.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c: continue
    dur = c["GRBM_GUI_ACTIVE"] / 8.0                     # summed over 8 XCDs -> cycles
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0        # summed over 1024 SIMDs -> cycles per SIMD
    tot[0] += dur; tot[1] += busy
    f = family(name[k])
    if f:
        a = agg[f]; a[0] += dur; a[1] += busy; a[2] += 1
print("# matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), kernels of")
print("# bench.py --steps 3 --warmup 2 (5 steps + start-up), every launch counted")
print(f"{'family':58s} {'launches':>8s} {'share of GPU cycles':>20s} {'matrix pipe busy':>17s}")
for f, (d, b, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{f:58s} {n:8d} {100 * d / tot[0]:19.1f}% {100 * b / d:16.1f}%")
print(f"{'all kernels of the run':58s} {len(per):8d} {100.0:19.1f}% {100 * tot[1] / tot[0]:16.1f}%")
PY
cat "$OUT"
