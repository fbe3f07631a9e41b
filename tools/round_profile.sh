# The round's measurements on one GPU box: GPU tests, bench lines (all stages), steady-state kernel table,
# in-step matrix-pipe counters, MSDA HBM counters.  usage (GPU box): bash tools/round_profile.sh <outdir>
OUT=${1:-gpurun_out/r4m}
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -3 $OUT/gputests.log
python bench.py --steps 50 --warmup 20 > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log > $OUT/bench_line.json; cut -c1-400 $OUT/bench_line.json
python bench.py --stage source-only --steps 20 --warmup 8 > $OUT/bench_src.log 2>&1; tail -1 $OUT/bench_src.log > $OUT/bench_source_only_line.json; cut -c1-300 $OUT/bench_source_only_line.json
python bench.py --stage teacher --steps 10 --warmup 4 > $OUT/bench_teacher.log 2>&1; tail -1 $OUT/bench_teacher.log > $OUT/bench_teacher_line.json; cut -c1-300 $OUT/bench_teacher_line.json
DATR_DIST_FORCE_COLLECTIVES=1 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --trained-like-steps 0 > $OUT/bench_rccl1.log 2>&1; tail -1 $OUT/bench_rccl1.log > $OUT/bench_line_one_rank_rccl.json; cut -c1-300 $OUT/bench_line_one_rank_rccl.json
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/kstats.py $f --steps 4 --marker msda_fwd_pyr2 --per-step 6 --out $OUT/step_kernels.csv --top 140 --split msda_fwd_pyr2 --split-out $OUT/msda_fwd_by_grid.csv 2>&1 | tail -4
python tools/kstats.py $f --steps 4 --marker msda_fwd_pyr2 --per-step 6 --top 1 --split gemm_f32_kernel --split-out $OUT/gemm_by_grid.csv > /dev/null 2>&1
python tools/kfamilies.py $OUT/step_kernels.csv > $OUT/step_families.txt 2>&1
s=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); head -41 $s > $OUT/rocprof_kernel_stats_top40.csv
bash tools/pmc_step_mfma.sh $OUT/step_mfma.txt > /dev/null 2>&1; tail -3 $OUT/step_mfma.txt
bash tools/pmc_msda_raw.sh $OUT/msda_pmc.json 2>&1 | tail -3
python tools/probes/gemm_calls.py 2>/dev/null | grep -v "^\[gpurun\]" > $OUT/library_gemm_calls.txt; head -3 $OUT/library_gemm_calls.txt
python tools/probes/aten_dispatch_sources.py --rows 200 2>/dev/null > $OUT/aten_dispatch.txt; grep "device ATen ops" $OUT/aten_dispatch.txt
python tools/probes/wino_layers.py 2>/dev/null > $OUT/wino_layers.txt; tail -2 $OUT/wino_layers.txt
