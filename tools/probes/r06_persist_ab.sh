#!/bin/bash
# Round 6: persistent workgroups in the phased pyramid kernels (forward + backward dots): parity, then kernel durations
cd "$(dirname "$0")/../.."
[ "$1" = notest ] || python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -3
for d in model gauss2.5; do for p in 0 1 2 3; do
  echo "== $d DATR_MSDA_PYR2_PERSIST=$p"
  DATR_MSDA_PYR2_PERSIST=$p bash tools/probes/kernel_times.sh 6 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 30 --envelope measured | grep -E "pyr2|bwd_pyr" | cut -c1-40,60-
done; done
