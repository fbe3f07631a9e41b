#!/bin/bash
# The backward's LDS-window kernel (msda_bwd_dots_pyr2_d32) under compile-time switches (-D flags of csrc/msda_fwd_pyr2.hip), timed
# under rocprofv3.   P2_FLAGSETS="name:-DX=1,-DY=2 name2:..."
#   bash tools/probes/dots_flags.sh build   (here)      bash tools/probes/dots_flags.sh run   (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
SETS="${P2_FLAGSETS:-d1:-DPYR2_DOTS_DEPTH=1 d2:-DPYR2_DOTS_DEPTH=2}"
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_fwd_pyr2)
  for fs in $SETS; do
    name=${fs%%:*}; flags=$(echo "${fs#*:}" | tr ',' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $flags -c $C/msda_fwd_pyr2.hip -o /tmp/p2_f$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_p2f_$name.so $OTHERS /tmp/p2_f$name.o
  done
else
  for fs in $SETS; do
    name=${fs%%:*}
    for d in ${P2_DISTS:-model}; do
    echo -n "$name dist=$d "
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_p2f_$name.so bash tools/probes/kernel_times.sh 3 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 20 --envelope measured | grep "msda_" | cut -c28-52,98-
    done
  done
fi
