# MSDA backward (encoder call, N = 4) under window halos: envelope-sized (product), widest that fits, fixed
for d in model gauss1.5 gauss2.5; do
echo "== dist $d"
for env in measured none; do
echo -n "envelope=$env "; python tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 20 --envelope $env 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd', d['bwd_us_median'], d['bwd_us_min'], 'fwd', d['fwd_us_median'])"
done; done
