#!/bin/bash
# Launch-shape variants of the backward's sorted-scatter kernel (csrc/msda_bwd_pyr.hip: PYRB_THREADS,
# PYRB_MAXQ, PYRB_INFLIGHT), timed under rocprofv3 on the N=4 encoder call.
#   bash tools/probes/bwd_scatter_cfgs.sh build   (here)      bash tools/probes/bwd_scatter_cfgs.sh run   (GPU box)
set -e
cd "$(dirname "$0")/../.."
C=datr_amd/csrc
VARIANTS="${PYRB_VARIANTS:-512,256,8 512,256,4 512,256,16 256,128,8 768,384,8 1024,512,8 384,192,8}"
if [ "$1" = build ]; then
  make -C $C >/dev/null
  OTHERS=$(ls $C/build/*.o | grep -v msda_bwd_pyr)
  for v in $VARIANTS; do
    IFS=, read t q f <<< "$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DPYRB_THREADS=$t -DPYRB_MAXQ=$q -DPYRB_INFLIGHT=$f $PYRB_FLAGS -c $C/msda_bwd_pyr.hip -o /tmp/pyrb_$t$q$f.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o datr_amd/lib/libdatr_hip_pb_${t}_${q}_$f.so $OTHERS /tmp/pyrb_$t$q$f.o
  done
else
  for v in $VARIANTS; do
    IFS=, read t q f <<< "$v"
    echo -n "threads=$t maxq=$q inflight=$f "
    DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_pb_${t}_${q}_$f.so bash tools/probes/kernel_times.sh 3 python $PWD/tools/bench_msda.py --dist ${PYR2_DIST:-model} --n 4 --encoder-only --iters 20 --envelope measured | grep "bwd_pyr_d32" | cut -c60-
  done
fi
