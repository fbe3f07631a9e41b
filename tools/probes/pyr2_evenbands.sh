export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" datr_amd/lib/libdatr_hip_prev.so; do
  echo "== lib=${lib:-default}"
  DATR_HIP_LIB=$lib python tools/bench_msda.py --dist model --n 4 --fwd-only --encoder-only --iters 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fwd_us_median'], d['fwd_us_min'])"
done
done
python -m pytest tests/test_msda_gpu.py -q -x 2>&1 | tail -2
