#!/bin/bash
# counters of the round-6 visits scatter and of the corner-record scatter it replaces (same inputs)
cd /root/repo
export TMPDIR=/tmp
DIST=${1:-model}
tools/pmc_kernel.sh msda_bwd_visits gpurun_out/r06_bwd_visits_pmc_$DIST.txt -- python tools/bench_msda.py --dist $DIST --n 4 --encoder-only --iters 6 > /dev/null 2>&1
DATR_MSDA_BWD_VISITS=0 tools/pmc_kernel.sh msda_bwd_pyr_d32 gpurun_out/r06_bwd_records_pmc_$DIST.txt -- python tools/bench_msda.py --dist $DIST --n 4 --encoder-only --iters 6 > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o v -- python /root/repo/tools/bench_msda.py --dist $DIST --n 4 --encoder-only --iters 20 > /dev/null 2>&1
f=$(find /tmp/prof_v -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-200 > /root/repo/gpurun_out/r06_bwd_visits_kstats_$DIST.txt
