cd $GRAFT_REPO_ROOT
python -m pytest tests/test_msda_gpu.py -x -q 2>&1 | tail -4
for d in model gauss1.5 gauss2.5; do
for lib in libdatr_hip_r4bwd.so libdatr_hip.so; do
  echo "== $lib dist=$d"
  DATR_HIP_LIB=$PWD/datr_amd/lib/$lib bash tools/probes/kernel_times.sh 4 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 20 --envelope measured | grep -i "bwd" | cut -c1-130
done; done
