#!/bin/bash
# LDS activity of the MSDA kernels from rocprofv3 PMC counters (counter passes only, no trace domains
# other than --kernel-trace).   usage: tools/pmc_lds.sh <outdir>   (run on the GPU box)
set -e
OUT=${1:-gpurun_out/pmc_lds}
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/pmcl
for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS; do
  rm -rf /tmp/pmcl/$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcl/$C -o run -- \
      python tools/bench_msda.py --iters 2 --dist model > /tmp/pmcl/$C.log 2>&1 || true
  f=$(find /tmp/pmcl/$C -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "$C: no output"; continue; }
  python - "$f" "$C" <<'PY' | tee "$OUT/$C.txt"
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != name: continue
    k = r["Kernel_Name"]
    if "msda" not in k: continue
    agg[(k[:48], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for (k, g), v in sorted(agg.items()):
    print(f"{name} kernel={k} grid={g} launches={len(v)} mean={sum(v)/len(v):.4g}")
PY
done
