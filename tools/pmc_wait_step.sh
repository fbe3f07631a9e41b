#!/bin/bash
# Per kernel of the training step: how much of the waves' life issues instructions, waits, and how many waves there are
# (one PMC pass).  usage (GPU box): bash tools/pmc_wait_step.sh
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pw; mkdir -p /tmp/pw
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pw -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 > /tmp/pw.log 2>&1
f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:52]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:45]
print(f"{'kernel':52s} {'launches':>8s} {'busy cyc':>10s} {'waves/launch':>12s} {'issue':>6s} {'wait':>6s} {'valu':>6s}")
for k, c in rows:
    wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k:52s} {n[k]:8d} {c.get('SQ_BUSY_CYCLES', 0):10.3g} {c.get('SQ_WAVES', 0) / max(n[k], 1):12.0f} "
          f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.2f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:6.2f} {c.get('SQ_ACTIVE_INST_VALU', 0) / wc:6.2f}")
PY
