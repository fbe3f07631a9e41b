mkdir -p gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --padded-steps 0 > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/kstats.py $f --steps 4 --marker msda_fwd_pyr2 --per-step 6 --out gpurun_out/r3f/step_kernels.csv --top 120 2>&1 | tail -4
python tools/probes/elementwise_audit.py $f 4 > gpurun_out/r3f/elementwise_audit.txt 2>&1
python tools/probes/idle_gaps.py $f 4 > gpurun_out/r3f/idle_gaps.txt 2>&1
