cd /root/repo
for rep in 1 2 3; do for d in model gauss2.5; do for v in passes queue; do
  echo -n "$rep $d $v  "
  DATR_HIP_LIB=$PWD/datr_amd/lib/libdatr_hip_f_$v.so bash tools/probes/kernel_times.sh 6 python $PWD/tools/bench_msda.py --dist $d --n 4 --encoder-only --iters 40 --envelope measured | grep -E "bwd_dot" | cut -c60-
done; done; done
