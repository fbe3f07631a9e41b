#!/bin/bash
# Builds ablated variants of the phased pyramid MSDA forward (csrc/msda_fwd_pyr2.hip, PYR2_ABLATE
# bits: 1 no window fill, 4 no LDS gathers, 8 no loc/attn loads, 16 no output stores) next to the
# production library and times each on the N=4 encoder call.
#   bash tools/probes/pyr2_ablate.sh build      (here: hipcc cross-compiles)
#   bash tools/probes/pyr2_ablate.sh run        (on the GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
VARIANTS="${PYR2_VARIANTS:-0 1 4 8 16 5 12 13 29}"
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_fwd_pyr2)
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DPYR2_ABLATE=$v $PYR2_FLAGS -c $C/msda_fwd_pyr2.hip -o /tmp/pyr2_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_p2a$v$PYR2_TAG.so $OTHERS /tmp/pyr2_$v.o
  done
else
  for v in $VARIANTS; do
    echo -n "ablate=$v$PYR2_TAG "
    DATR_HIP_LIB=datr_amd/lib/libdatr_hip_p2a$v$PYR2_TAG.so python tools/bench_msda.py --dist ${PYR2_DIST:-model} --n 4 --fwd-only --encoder-only --iters 30 2>&1 | grep 22223 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fwd_us_median'], d['fwd_us_min'])"
  done
fi
