for d in model gauss1.5 gauss2.5; do for c in 4.75 5.5 24; do
echo -n "dist $d clip $c: "; DATR_MSDA_PYR2_ENV_CLIP=$c python tools/bench_msda.py --dist $d --n 4 --encoder-only --fwd-only --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd', d['fwd_us_median'], d['fwd_us_min'])"
done; done
