#!/bin/bash
# MFMA / LDS / wave-cycle counters of the own MFMA kernels (attention forward, 256x256 weight
# gradient) from rocprofv3 PMC passes (one counter per pass, --kernel-trace only).
#   usage: tools/pmc_mfma.sh <outdir>   (run on the GPU box)
set -e
OUT=${1:-gpurun_out/pmc_mfma}
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/pmcm
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE; do
  for P in tools/probes/mha_probe.py; do
    T=$(basename $P .py)
    rm -rf /tmp/pmcm/$C.$T
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcm/$C.$T -o run -- \
        python $P > /tmp/pmcm/$C.$T.log 2>&1 || true
    f=$(find /tmp/pmcm/$C.$T -name "*counter_collection.csv" | head -1)
    [ -z "$f" ] && { echo "$C $T: no output"; continue; }
    python - "$f" "$C" <<'PY' | tee -a "$OUT/summary.txt"
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != name: continue
    k = r["Kernel_Name"]
    if "mha_fwd" not in k: continue
    agg[k[:60]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{name} kernel={k} launches={len(v)} mean={sum(v)/len(v):.5g}")
PY
  done
done
