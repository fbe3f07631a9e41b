#!/bin/bash
# Round 6: region grids of the phased pyramid forward on the N = 4 encoder call (rocprofv3 kernel durations)
cd "$(dirname "$0")/../.."
for d in ${PYR2_DISTS:-model gauss2.5}; do
for g in default 8x12 6x16 7x14 8x14 7x12 6x14 8x16 12x8 4x24; do
  echo -n "$d grid=$g "
  if [ $g = default ]; then unset DATR_MSDA_PYR2_REGIONS; else export DATR_MSDA_PYR2_REGIONS=$g; fi
  bash tools/probes/kernel_times.sh 4 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 30 --envelope measured --fwd-only | grep "fwd_pyr2" | cut -c60-
done; done
