#!/bin/bash
# HBM traffic of the MSDA kernels from rocprofv3 PMC counters (separate passes per counter, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; never combined with trace domains).
# usage: tools/pmc_msda.sh <outdir>     (run on the GPU box)
set -e
OUT=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc/$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc/$C -o run -- \
      python tools/bench_msda.py --iters 3 --dist model > /tmp/pmc/$C.log 2>&1 || true
  f=$(find /tmp/pmc/$C -name "*counter_collection.csv" | head -1)
  python - "$f" "$C" <<'PY' | tee "$OUT/$C.txt"
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != name: continue
    k = r["Kernel_Name"]
    if "msda" not in k: continue
    agg[(k[:60], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for (k, g), v in sorted(agg.items()):
    print(f"{name} kernel={k} grid={g} launches={len(v)} mean={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
PY
done
