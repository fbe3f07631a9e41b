#!/bin/bash
# The round-6 visits scatter with pieces compiled out / other shapes, timed under rocprofv3.
#   bash tools/probes/r06_bwd_visits_ablate.sh build   (build container)     ... run   (GPU box)
# variants: "<name>:<extra -D flags>"
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
VARIANTS=${PYRV_VARIANTS:-"base: noflush:-DPYRV_ABLATE=1 noloop:-DPYRV_ABLATE=2 noloopflush:-DPYRV_ABLATE=3 nosort:-DPYRV_ABLATE=12 fly8:-DPYRV_INFLIGHT=8 fly2:-DPYRV_INFLIGHT=2 seg4:-DPYRV_SEG=4 seg1:-DPYRV_SEG=1 occ4:-DPYRB_WAVES_PER_SIMD=4"}
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_bwd_pyr)
  for v in $VARIANTS; do
    n=${v%%:*}; f=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $f -c $C/msda_bwd_pyr.hip -o /tmp/pyrv_$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_pv_$n.so $OTHERS /tmp/pyrv_$n.o
  done
else
  for v in $VARIANTS; do
    n=${v%%:*}
    echo -n "$n "
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_pv_$n.so bash tools/probes/kernel_times.sh 4 python $PWD/tools/bench_msda.py --dist ${PYR2_DIST:-model} --n 4 --encoder-only --iters 20 --envelope measured | grep "bwd_visits" | cut -c60-
  done
fi
