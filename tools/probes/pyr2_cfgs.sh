# times the configuration builds of the phased pyramid forward (datr_amd/lib/libdatr_hip_cfg*.so)
for t in ${CFGS:-a b}; do
  echo "== cfg $t"
  DATR_HIP_LIB=datr_amd/lib/libdatr_hip_cfg$t.so python tools/bench_msda.py --dist ${DIST:-model} --n 4 --fwd-only --encoder-only --iters 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['plan']['grid'], d['plan']['tasks_per_wave'], d['fwd_us_median'], d['fwd_us_min'])"
  PYR2_PROBE=1 DATR_HIP_LIB=datr_amd/lib/libdatr_hip_cfg${t}_prb.so python tools/bench_msda.py --dist ${DIST:-model} --n 4 --fwd-only --encoder-only --iters 5 2>&1 | grep -E "phase cycles|wg spans"
done
