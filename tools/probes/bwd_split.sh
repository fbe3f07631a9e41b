# MSDA encoder backward (N = 4): grad_loc / grad_attn by the LDS-window kernel (product) vs the scatter kernel's own gathers
for d in model gauss1.5 gauss2.5; do
for split in 1 0; do
echo -n "dist=$d split=$split "; DATR_MSDA_BWD_SPLIT=$split python tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 20 --envelope measured 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd', d['bwd_us_median'], d['bwd_us_min'], 'fwd', d['fwd_us_median'])"
done; done
