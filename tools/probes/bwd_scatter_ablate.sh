#!/bin/bash
# The backward's sorted-scatter kernel with pieces compiled out (csrc/msda_bwd_pyr.hip PYRB_ABLATE bits:
# 1 no float-atomic flush, 2 no reduce loop, 4 no records, 8 no histogram atomics), timed under rocprofv3.
#   bash tools/probes/bwd_scatter_ablate.sh build   (here)      bash tools/probes/bwd_scatter_ablate.sh run   (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
VARIANTS="${PYRB_VARIANTS:-0 1 2 3 4 12 15}"
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_bwd_pyr)
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DPYRB_ABLATE=$v -c $C/msda_bwd_pyr.hip -o /tmp/pyrb_a$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_pba$v.so $OTHERS /tmp/pyrb_a$v.o
  done
else
  for v in $VARIANTS; do
    echo -n "ablate=$v "
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_pba$v.so bash tools/probes/kernel_times.sh 3 python $PWD/tools/bench_msda.py --dist ${PYR2_DIST:-model} --n 4 --encoder-only --iters 20 --envelope measured | grep "bwd_pyr_d32" | cut -c60-
  done
fi
