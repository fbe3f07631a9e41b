#!/bin/bash
# Matrix-pipe busy of the own GEMM family's launches of the training step, by template instance and grid size
# (one PMC pass).  usage (GPU box): bash tools/pmc_gemm_mfma_by_grid.sh
export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pg
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pg -o run -- python $REPO/bench.py --steps 2 --warmup 2 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 > /tmp/pg.log 2>&1
python - "$(find /tmp/pg -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
per = collections.defaultdict(dict); meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Dispatch_Id"]; per[k][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[k] = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44], r.get("Grid_Size", ""))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k, c in per.items():
    n, g = meta[k]
    if "gemm_f32_kernel" not in n: continue
    a = agg[(n, g)]; a[0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); a[1] += c.get("GRBM_GUI_ACTIVE", 0); a[2] += 1
tot = sum(a[1] for a in agg.values())
print(f"{'instance':44s} {'grid':>9s} {'launches':>8s} {'share of family cycles':>22s} {'matrix pipe busy':>17s}")
for (n, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:44s} {g:>9s} {a[2]:8d} {a[1] / tot:22.3f} {a[0] / 1024 / (a[1] / 8):17.3f}")
PY
