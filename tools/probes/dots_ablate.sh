#!/bin/bash
# Ablated variants of the backward's LDS-window kernel (msda_bwd_dots_pyr2_d32 in csrc/msda_fwd_pyr2.hip;
# PYR2_ABLATE bits as in pyr2_ablate.sh + 256 no grad stores, 512 no quad sums), timed under rocprofv3.
#   bash tools/probes/dots_ablate.sh build   (here)      bash tools/probes/dots_ablate.sh run   (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
VARIANTS="${PYR2_VARIANTS:-0 256 512 768 4 1 8}"
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_fwd_pyr2)
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DPYR2_ABLATE=$v $PYR2_FLAGS -c $C/msda_fwd_pyr2.hip -o /tmp/pyr2_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_p2a$v.so $OTHERS /tmp/pyr2_$v.o
  done
else
  for v in $VARIANTS; do
    echo -n "ablate=$v "
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_p2a$v.so bash tools/probes/kernel_times.sh 3 python $PWD/tools/bench_msda.py --dist ${PYR2_DIST:-model} --n 4 --encoder-only --iters 20 --envelope measured | grep dots
  done
fi
