export TMPDIR=/tmp
for lib in ${LIBS:-"" datr_amd/lib/libdatr_hip_bbands.so}; do
  echo "== lib=${lib:-default}"
  DATR_HIP_LIB=$lib python tools/bench_msda.py --dist model --n 4 --encoder-only --iters 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd', d['bwd_us_median'], d['bwd_us_min'], 'fwd', d['fwd_us_median'])"
  for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pb; (cd /tmp && DATR_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pb -o run -- python $GRAFT_REPO_ROOT/tools/bench_msda.py --iters 4 --dist model --n 4 --encoder-only > /tmp/pb.log 2>&1)
  python - $C <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/pb/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "msda_bwd_pyr" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[1]]
print(sys.argv[1], "KB mean", sum(v) / max(1, len(v)), "launches", len(v))
PY
  done
done
