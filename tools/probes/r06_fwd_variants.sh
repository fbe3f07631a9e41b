#!/bin/bash
# The phased pyramid forward (csrc/msda_fwd_pyr2.hip) built with other -D flags, timed under rocprofv3.
#   bash tools/probes/r06_fwd_variants.sh build   (build container)     ... run   (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
VARIANTS=${PYR2_VARIANTS:-"base: fmacdpp:-DPYR2_FMAC_DPP=1"}
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_fwd_pyr2)
  for v in $VARIANTS; do
    n=${v%%:*}; f=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude $f -c $C/msda_fwd_pyr2.hip -o /tmp/pyr2_$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_f_$n.so $OTHERS /tmp/pyr2_$n.o
  done
else
  for d in ${PYR2_DISTS:-model gauss2.5}; do
  for v in $VARIANTS; do
    n=${v%%:*}
    echo -n "$d $n "
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_f_$n.so bash tools/probes/kernel_times.sh 4 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 30 --envelope measured --fwd-only | grep "fwd_pyr2" | cut -c60-
  done; done
  if [ -n "$PYR2_TEST" ]; then
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_f_$PYR2_TEST.so python -m pytest tests/test_msda_gpu.py -x -q -m gpu -k "forward or fwd or pyramid" 2>&1 | tail -3
  fi
fi
