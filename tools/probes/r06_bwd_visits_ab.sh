#!/bin/bash
# A/B of the round-6 visits scatter against the corner-record scatter: parity tests, then HIP-event timings of the
# N = 4 encoder call (forward, backward = zero-fill + dots + scatter) for ring and N(0, 2.5 px) offsets
cd /root/repo
[ "$1" = "notest" ] || python -m pytest tests/test_msda_gpu.py -x -q -m gpu -k "backward or bwd or grad" 2>&1 | tail -3
for v in 1 0; do for d in model gauss2.5; do
DATR_MSDA_BWD_VISITS=$v python tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VISITS=$v', d['dist'], 'fwd', d['fwd_us_median'], 'bwd', d['bwd_us_median'])"
done; done
