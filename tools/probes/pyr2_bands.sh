export TMPDIR=/tmp
for lib in ${LIBS:-"" datr_amd/lib/libdatr_hip_bands.so}; do
  echo "== lib=${lib:-default}"
  DATR_HIP_LIB=$lib python tools/bench_msda.py --dist model --n 4 --fwd-only --encoder-only --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fwd_us_median'], d['fwd_us_min'])"
  rm -rf /tmp/pb; (cd /tmp && DATR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pb -o run -- python $GRAFT_REPO_ROOT/tools/bench_msda.py --iters 4 --dist model --n 4 --encoder-only --fwd-only > /tmp/pb.log 2>&1)
  python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pb/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "msda_fwd_pyr2" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("FETCH_SIZE KB mean", sum(v) / max(1, len(v)), "launches", len(v))
PY
done
