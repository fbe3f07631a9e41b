# steady-state kernel table of the training step (rocprofv3 kernel trace of bench.py, last 4 steps)
#   usage (GPU box): bash tools/step_trace.sh <outdir>
OUT=${1:-gpurun_out/r4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --padded-steps 0 --trained-like-steps 0 > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/kstats.py $f --steps 4 --marker msda_fwd_pyr2 --per-step 6 --out $OUT/step_kernels.csv --top 140 --split gemm_f32_kernel --split-out $OUT/gemm_by_grid.csv 2>&1 | tail -4
python tools/probes/elementwise_audit.py $f 4 > $OUT/elementwise_audit.txt 2>&1
python tools/probes/idle_gaps.py $f 4 > $OUT/idle_gaps.txt 2>&1
