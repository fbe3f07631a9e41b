mkdir -p gpurun_out/r3m
python -m pytest tests -m gpu -q > gpurun_out/r3m/gputests.log 2>&1; tail -3 gpurun_out/r3m/gputests.log
python bench.py --steps 50 --warmup 20 > gpurun_out/r3m/bench.log 2>&1; tail -1 gpurun_out/r3m/bench.log > gpurun_out/r3m/bench_line.json; cut -c1-400 gpurun_out/r3m/bench_line.json
python bench.py --stage source-only --steps 20 --warmup 8 > gpurun_out/r3m/bench_src.log 2>&1; tail -1 gpurun_out/r3m/bench_src.log > gpurun_out/r3m/bench_source_only_line.json; cut -c1-300 gpurun_out/r3m/bench_source_only_line.json
python bench.py --stage teacher --steps 10 --warmup 4 > gpurun_out/r3m/bench_teacher.log 2>&1; tail -1 gpurun_out/r3m/bench_teacher.log > gpurun_out/r3m/bench_teacher_line.json; cut -c1-300 gpurun_out/r3m/bench_teacher_line.json
DATR_DIST_FORCE_COLLECTIVES=1 python bench.py --steps 20 --warmup 8 --no-cpu-baseline > gpurun_out/r3m/bench_rccl1.log 2>&1; tail -1 gpurun_out/r3m/bench_rccl1.log > gpurun_out/r3m/bench_line_one_rank_rccl.json; cut -c1-300 gpurun_out/r3m/bench_line_one_rank_rccl.json
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/kstats.py $f --steps 4 --marker msda_fwd_pyr2 --per-step 6 --out gpurun_out/r3m/step_kernels.csv --top 90 --split msda_fwd_pyr2 --split-out gpurun_out/r3m/msda_fwd_by_grid.csv 2>&1 | tail -4
s=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); head -41 $s > gpurun_out/r3m/rocprof_kernel_stats_top40.csv
bash tools/pmc_step_mfma.sh gpurun_out/r3m/step_mfma.txt > /dev/null 2>&1; tail -3 gpurun_out/r3m/step_mfma.txt
bash tools/pmc_msda_raw.sh gpurun_out/r3m/msda_pmc.json 2>&1 | tail -3
