#!/bin/bash
# Builds ablated variants of the pyramid-region MSDA forward (csrc/msda_fwd_pyr.hip, PYR_ABLATE
# bits) next to the production library and times each on the N=4 encoder call.
#   bash tools/probes/pyr_ablate.sh build      (here: hipcc cross-compiles)
#   bash tools/probes/pyr_ablate.sh run        (on the GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
VARIANTS="${PYR_VARIANTS:-0 1 2 4 6 8 16 31}"
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_fwd_pyr)
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DPYR_ABLATE=$v $PYR_FLAGS -c $C/msda_fwd_pyr.hip -o /tmp/pyr_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_pyr$v$PYR_TAG.so $OTHERS /tmp/pyr_$v.o
  done
else
  for v in $VARIANTS; do
    echo -n "ablate=$v "
    DATR_HIP_LIB=datr_amd/lib/libdatr_hip_pyr$v$PYR_TAG.so python tools/bench_msda.py --dist ${PYR_DIST:-model} --n 4 --fwd-only --iters 30 2>&1 | grep 22223 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fwd_us_median'], d['fwd_us_min'])"
  done
fi
