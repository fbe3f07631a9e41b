#!/bin/bash
# LDS conflict share per kernel over the training step (two PMC passes)
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pl; mkdir -p /tmp/pl
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pl -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 > /tmp/pl.log 2>&1
f=$(find /tmp/pl -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    k = k.replace("(anonymous namespace)::", "")[:46]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:22]
print(f"{'kernel':46s} {'LDS active':>12s} {'conflict':>12s} {'share':>6s} {'wait LDS / wave cycles':>24s}")
for k, c in rows:
    a, b = c.get("SQ_LDS_IDX_ACTIVE", 0), c.get("SQ_LDS_BANK_CONFLICT", 0)
    w, wc = c.get("SQ_WAIT_INST_LDS", 0), c.get("SQ_WAVE_CYCLES", 1)
    print(f"{k:46s} {a:12.3g} {b:12.3g} {b / max(a, 1):6.2f} {w / wc:24.3f}")
PY
