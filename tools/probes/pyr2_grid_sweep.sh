for g in "" 5x12 7x14 8x12 10x16 12x16 16x16; do
  echo -n "grid ${g:-planner}: "
  DATR_MSDA_PYR2=2 DATR_MSDA_PYR2_REGIONS=$g python tools/bench_msda.py --dist model --n 4 --fwd-only --encoder-only --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plan']['grid'], d['plan']['phases'], d['plan']['tasks_per_wave'], d['plan']['fill_kib'], d['fwd_us_median'], d['fwd_us_min'])"
done
for d in gauss1.5 gauss2.5; do for p2 in 1 2 0; do echo -n "$d PYR2=$p2: "; DATR_MSDA_PYR2=$p2 python tools/bench_msda.py --dist $d --n 4 --fwd-only --encoder-only --iters 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plan']['grid'], d['plan']['phases'], d['plan']['phased'], d['fwd_us_median'], d['fwd_us_min'])"; done; done
echo -n "model, no envelope, PYR2=2: "; DATR_MSDA_PYR2=2 python tools/bench_msda.py --dist model --n 4 --fwd-only --encoder-only --envelope none --iters 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plan']['grid'], d['plan']['phases'], d['fwd_us_median'], d['fwd_us_min'])"
