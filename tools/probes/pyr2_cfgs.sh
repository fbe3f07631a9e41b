# times the configuration builds of the phased pyramid forward (datr_amd/lib/libdatr_hip_cfg*.so)
for t in ${CFGS:-default d2 t256 w3}; do
  lib=datr_amd/lib/libdatr_hip_cfg$t.so; [ $t = default ] && lib=""
  echo -n "cfg $t: "
  DATR_HIP_LIB=$lib python tools/bench_msda.py --dist ${DIST:-model} --n 4 --fwd-only --encoder-only --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plan']['grid'], d['plan']['phases'], d['plan']['tasks_per_wave'], d['fwd_us_median'], d['fwd_us_min'])"
done
