#!/bin/bash
# HBM traffic of the encoder MSDA kernels as RAW per-launch rows (rocprofv3 PMC, one counter per
# pass with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
# Writes <out>.json with the FETCH_SIZE / WRITE_SIZE value (KB) of every launch of the N = 4 encoder
# forward (the kernel bench.py's roofline names) and backward; bench.py applies
#   bytes = (2 x mean FETCH_SIZE + mean WRITE_SIZE) x 1024
# itself (gfx950 tallies 128-B requests of 16 B/lane reads at 64 B: FETCH_SIZE x 2; WRITE_SIZE as reported).
#   usage: tools/pmc_msda_raw.sh profiles/r03_msda_pmc.json      (on the GPU box, from the repo root)
set -e
OUT=${1:-gpurun_out/msda_pmc.json}
REPO=$PWD
export TMPDIR=/tmp
mkdir -p "$(dirname "$OUT")" /tmp/pmcraw
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcraw/$C
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmcraw/$C -o run -- \
      python $REPO/tools/bench_msda.py --iters 6 --dist model --n 4 --encoder-only > /tmp/pmcraw/$C.log 2>&1) || true
done
# pipe activity of the same launches (round 6: bench.py's `roofline.ceiling`): vector ALU, LDS data path, elapsed cycles
i=0
for P in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcraw/pipes$i
  (cd /tmp && rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcraw/pipes$i -o run -- \
      python $REPO/tools/bench_msda.py --iters 6 --dist model --n 4 --encoder-only > /tmp/pmcraw/pipes$i.log 2>&1) || true
done
python - "$OUT" <<'PY'
import csv, glob, json, re, sys
out = {"source": "tools/pmc_msda_raw.sh: rocprofv3 --pmc <counter> --kernel-trace, one counter per pass; "
                 "tools/bench_msda.py --dist model --n 4 --encoder-only (measured envelope)",
       "shape": {"N": 4, "S": 22223, "M": 8, "D": 32, "L": 4, "P": 4, "Lq": 22223},
       "unit": "KB as rocprofv3 reports FETCH_SIZE / WRITE_SIZE",
       "correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md: gfx950 tallies 128-B requests of 16 B/lane reads at 64 B); "
                     "WRITE_SIZE as reported; applied by the reader, the rows here are raw"}
kern = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmcraw/{c}/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    for r in rows:
        if r.get("Counter_Name") != c:
            continue
        k = r["Kernel_Name"]
        for tag, pat in (("forward", "msda_fwd_pyr"), ("backward", "msda_bwd_pyr"), ("backward_dots", "msda_bwd_dots")):
            if pat in k:
                kern.setdefault(tag, {"kernel": (re.search(r"msda_\w+(<\d+>)?", k) or [k])[0]}).setdefault(c + "_KB_per_launch", []).append(float(r["Counter_Value"]))
# pipe counters: mean per launch of the forward / backward kernels
pipes = {}
for f in glob.glob("/tmp/pmcraw/pipes*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for tag, pat in (("forward", "msda_fwd_pyr"), ("backward", "msda_bwd_pyr"), ("backward_dots", "msda_bwd_dots")):
            if pat in k:
                pipes.setdefault(tag, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for tag, d in pipes.items():
    kern.setdefault(tag, {})["pipe_counters_mean_per_launch"] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
out["pipe_counter_units"] = ("SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: quad-cycles summed over the SIMDs (x 4 = cycles; 1024 SIMDs); "
                             "SQ_LDS_IDX_ACTIVE: LDS-array cycles summed over the 256 CUs; GRBM_GUI_ACTIVE: cycles summed over the 8 XCDs")
fw = kern.get("forward", {})
out["kernel"] = fw.get("kernel")
out["pipe_counters_mean_per_launch"] = fw.get("pipe_counters_mean_per_launch", {})
out["FETCH_SIZE_KB_per_launch"] = fw.get("FETCH_SIZE_KB_per_launch", [])
out["WRITE_SIZE_KB_per_launch"] = fw.get("WRITE_SIZE_KB_per_launch", [])
out["backward"] = kern.get("backward", {})            # the sorted scatter (grad_value)
out["backward_dots"] = kern.get("backward_dots", {})  # grad_loc / grad_attn out of LDS windows
json.dump(out, open(sys.argv[1], "w"), indent=1)
for tag in ("forward", "backward", "backward_dots"):
    k = kern.get(tag, {})
    f_, w_ = k.get("FETCH_SIZE_KB_per_launch", []), k.get("WRITE_SIZE_KB_per_launch", [])
    if f_ and w_:
        print(tag, k.get("kernel"), "launches", len(f_), len(w_), "mean FETCH KB %.0f WRITE KB %.0f -> traffic %.1f MB" % (
            sum(f_) / len(f_), sum(w_) / len(w_), (2 * sum(f_) / len(f_) + sum(w_) / len(w_)) * 1024 / 1e6))
PY
