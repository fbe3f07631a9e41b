#!/bin/bash
# Round 6: per-quad miss queue in the pyramid kernels' slow path (PYR2_MISS_QUEUE=1, the source default) against the four
# fixed passes (=0): parity on the slow-path-heavy tests, then kernel durations for ring / N(0, 2.5 px) / N(0, 4 px) offsets.
#   bash tools/probes/r06_miss_queue_ab.sh build   (build container)      ... run   (GPU box)
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  PYR2_VARIANTS="queue:-DPYR2_MISS_QUEUE=1 passes:-DPYR2_MISS_QUEUE=0" bash tools/probes/r06_fwd_variants.sh build
else
  DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_f_queue.so python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -2
  for d in model gauss2.5 gauss4; do for v in passes queue; do
    echo "== $d $v"
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_f_$v.so bash tools/probes/kernel_times.sh 6 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 30 --envelope measured | grep -E "pyr2" | cut -c1-40,60-
  done; done
fi
