# region-grid sweep of the phased pyramid forward over the configuration builds
for t in ${CFGS:-b d e f}; do for g in ${GRIDS:-10x12 10x16 12x16 14x16 16x16 12x12}; do
  echo -n "cfg $t grid $g: "
  DATR_MSDA_PYR2_REGIONS=$g DATR_HIP_LIB=datr_amd/lib/libdatr_hip_cfg$t.so python tools/bench_msda.py --dist ${DIST:-model} --n 4 --fwd-only --encoder-only --iters 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plan']['forward'], d['plan']['phases'], d['plan']['tasks_per_wave'], d['plan']['fill_kib'], d['fwd_us_median'], d['fwd_us_min'])"
done; done
