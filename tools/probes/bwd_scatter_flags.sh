#!/bin/bash
# The backward's sorted-scatter kernel under compile-time switches (-D flags of csrc/msda_bwd_pyr.hip), timed
# under rocprofv3.   PYRB_FLAGSETS="name:-DX=1,-DY=2 name2:..."
#   bash tools/probes/bwd_scatter_flags.sh build   (here)      bash tools/probes/bwd_scatter_flags.sh run   (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
SETS="${PYRB_FLAGSETS:-base:-DPYRB_REC_DPP=0 dpp:-DPYRB_REC_DPP=1}"
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_bwd_pyr)
  for fs in $SETS; do
    name=${fs%%:*}; flags=$(echo "${fs#*:}" | tr ',' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags -c $C/msda_bwd_pyr.hip -o /tmp/pyrb_f$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_pbf_$name.so $OTHERS /tmp/pyrb_f$name.o
  done
else
  for fs in $SETS; do
    name=${fs%%:*}
    for d in ${PYRB_DISTS:-model}; do
    echo -n "$name dist=$d "
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_pbf_$name.so bash tools/probes/kernel_times.sh 3 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 20 --envelope measured | grep "bwd_pyr_d32" | cut -c60-
    done
  done
fi
