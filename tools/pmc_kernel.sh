#!/bin/bash
# SQ / LDS / TA counters of one kernel from rocprofv3 PMC passes (counter passes only, with
# --kernel-trace; never combined with other trace domains).  Run on the GPU box:
#   tools/pmc_kernel.sh <kernel-name-substring> <out.txt> -- <command ...>
set -e
PAT=$1; OUT=$2; shift 3
export TMPDIR=/tmp
mkdir -p "$(dirname "$OUT")" /tmp/pmck
: > "$OUT"
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmck/p$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmck/p$i -o run -- "$@" > /tmp/pmck/p$i.log 2>&1 || { echo "pass $i ($P) failed: $(tail -2 /tmp/pmck/p$i.log | tr '\n' ' ')" >> "$OUT"; continue; }
  f=$(find /tmp/pmck/p$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "pass $i ($P): no output" >> "$OUT"; continue; }
  python - "$f" "$PAT" >> "$OUT" <<'PY'
import csv, sys, collections
f, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if pat not in k: continue
    agg[(r["Counter_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for (c, g), v in sorted(agg.items()):
    print(f"{c:36s} grid={g:>9s} launches={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
done
cat "$OUT"
